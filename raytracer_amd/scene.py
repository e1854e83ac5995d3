"""Scene-file front end used by the harness (tests, bench, smoke).

Reads the reference's YAML scene language (src/scene.rs:229-382, :663-758, :910-1348) and
drives a :class:`raytracer_amd.builder.WorldBuilder` with the same construction order as
``Scene::render`` (src/scene.rs:73-213).  The loader is OUT of the accelerated path (SURVEY.md
§2 row 15): in the reference the Rust CLI keeps doing this; it exists here only because there
is no Rust toolchain to run the reference's own loader.

Numeric rules restated from the reference:
* expression strings (``PI/3``) are evaluated in f64 and cast to f32 (scene.rs:274-290, :1040-1043);
* plain numbers go f64 -> f32 (scene.rs:292-302);
* ``extend`` concatenates the base definitions' transform ops before the item's own ops and
  folds material specs in order (scene.rs:152-181, :629-661).
"""
from __future__ import annotations

import ast
import math
import os
from typing import Dict, List, Optional

import numpy as np
import yaml

from . import builder as B
from .obj import parse_obj

_CONSTS = {
    "PI": math.pi, "TAU": math.tau, "E": math.e, "FRAC_PI_2": math.pi / 2, "FRAC_PI_3": math.pi / 3,
    "FRAC_PI_4": math.pi / 4, "FRAC_PI_6": math.pi / 6, "FRAC_PI_8": math.pi / 8, "SQRT_2": math.sqrt(2.0),
}


def eval_expr(expr) -> float:
    """evalexpr 11 ``eval_float_with_context(expr, math_consts_context)`` for the forms the
    sample scenes use (constants, + - * / %, unary minus, parentheses).  int/int is integer
    division as in evalexpr; the result is an f64 that the caller casts to f32."""
    if isinstance(expr, (int, float)):
        return float(expr)
    node = ast.parse(str(expr).replace("^", "**"), mode="eval").body

    def ev(n):
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.Name) and n.id in _CONSTS:
            return _CONSTS[n.id]
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            v = ev(n.operand)
            return -v if isinstance(n.op, ast.USub) else v
        if isinstance(n, ast.BinOp):
            a, b = ev(n.left), ev(n.right)
            if isinstance(n.op, ast.Add):
                return a + b
            if isinstance(n.op, ast.Sub):
                return a - b
            if isinstance(n.op, ast.Mult):
                return a * b
            if isinstance(n.op, ast.Div):
                if isinstance(a, int) and isinstance(b, int):
                    return int(a / b)  # evalexpr integer division truncates
                return a / b
            if isinstance(n.op, ast.Mod):
                return math.fmod(a, b)
            if isinstance(n.op, ast.Pow):
                return float(a) ** float(b)
        raise ValueError(f"Invalid math expression: {expr!r}")

    return float(ev(node))


def parse_transform(spec) -> List[tuple]:
    """deser_transform (scene.rs:972-1143): list of [op, operands...] -> builder op tuples."""
    if not spec:
        return []
    ops = []
    for item in spec:
        name, args = item[0], item[1:]
        if name not in B.XF_KINDS:
            raise ValueError(f"Invalid transform operator: {name!r}")
        if name.startswith("rotate"):
            if len(args) != 1:
                raise ValueError(f"Invalid transform {name} operand: {args!r}")
            ops.append((name, eval_expr(args[0])))
        else:
            want = 6 if name == "shear" else 3
            if len(args) != want or any(isinstance(a, str) for a in args):
                raise ValueError(f"Invalid transform {name} operands: {args!r}")
            ops.append((name, *[float(a) for a in args]))
    return ops


_MATERIAL_FIELDS = ["color", "ambient", "diffuse", "specular", "shininess", "reflective", "transparency",
                    "refractive-index"]

_PATTERN_KINDS = {
    "striped": B.PAT_STRIPED, "gradient": B.PAT_GRADIENT, "ring": B.PAT_RING, "checker": B.PAT_CHECKER,
    "xyz-rgb": B.PAT_XYZ_RGB, "texture-map": B.PAT_TEXTURE_MAP, "cube-map": B.PAT_CUBE_MAP,
}
_UV_MAPPINGS = {"spherical": B.UVMAP_SPHERICAL, "planar": B.UVMAP_PLANAR, "cylindrical": B.UVMAP_CYLINDRICAL}
_CSG_OPS = {"union": B.CSG_UNION, "intersect": B.CSG_INTERSECT, "difference": B.CSG_DIFFERENCE}


def read_ppm(path: str) -> np.ndarray:
    """Canvas::from_ppm (src/canvas.rs:86-107): P3/P6 -> f32 RGB = sample / maxval."""
    with open(path, "rb") as f:
        data = f.read()
    # tokenise the header (magic, w, h, maxval), skipping comments
    pos, toks = 0, []
    while len(toks) < 4:
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while pos < len(data) and data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while pos < len(data) and not data[pos:pos + 1].isspace():
            pos += 1
        toks.append(data[start:pos])
    magic, w, h, maxval = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    if magic == b"P3":
        body = b" ".join(line.split(b"#")[0] for line in data[pos:].split(b"\n"))
        vals = np.array(body.split(), dtype=np.float64)[: w * h * 3]
    elif magic == b"P6":
        pos += 1
        dt = np.uint8 if maxval < 256 else ">u2"
        vals = np.frombuffer(data, dtype=dt, count=w * h * 3, offset=pos).astype(np.float64)
    else:
        raise ValueError(f"unsupported PNM magic {magic!r}")
    # the reference decodes to rgb8 first (into_rgb8) and then divides by maxval as f32
    return (vals.astype(np.float32) / np.float32(maxval)).reshape(h, w, 3)


class Scene:
    """A parsed scene file: ``Scene::render`` minus the rendering (src/scene.rs:73-213)."""

    def __init__(self, instructions: list, obj_files: Optional[Dict[str, str]] = None,
                 ppm_files: Optional[Dict[str, str]] = None, meshes: Optional[dict] = None):
        self.instructions = instructions
        self.obj_files = dict(obj_files or {})
        self.ppm_files = dict(ppm_files or {})
        self.meshes = dict(meshes or {})  # name -> pre-parsed mesh (list of OBJ groups), e.g. synthetic
        if not any(isinstance(i, dict) and i.get("add") == "camera" for i in instructions):
            raise ValueError("Missing camera")
        if not any(isinstance(i, dict) and i.get("add") in ("point-light", "area-light") for i in instructions):
            raise ValueError("Missing light")

    @classmethod
    def from_file(cls, path: str, obj_files=(), ppm_files=(), meshes=None) -> "Scene":
        with open(path) as f:
            instructions = yaml.safe_load(f)
        return cls(instructions, {os.path.basename(p): p for p in obj_files},
                   {os.path.basename(p): p for p in ppm_files}, meshes)

    @classmethod
    def from_text(cls, text: str, **kw) -> "Scene":
        return cls(yaml.safe_load(text), **kw)

    # ------------------------------------------------------------------------------------
    def camera(self, width: Optional[int] = None, height: Optional[int] = None) -> B.Camera:
        cam = None
        for ins in self.instructions:
            if ins.get("add") == "camera":
                cam = B.Camera(int(ins["width"]), int(ins["height"]), eval_expr(ins["field-of-view"]),
                               from_=ins["from"], to=ins["to"], up=ins["up"])
        if width is not None:
            cam.hsize, cam.vsize = int(width), int(height)
        return cam

    # ------------------------------------------------------------------------------------
    def build(self, wb: B.WorldBuilder) -> B.WorldBuilder:
        self._wb = wb
        self._define_transforms: Dict[str, list] = {}
        self._define_materials: Dict[str, list] = {}
        self._images: Dict[str, int] = {}
        self._mesh_cache: Dict[str, list] = {}
        # On the product path (a HostWorld) files go through the native loaders of include/rt_loaders.h; the oracle's
        # world is fed by the Python readers of this package (tests/test_loaders.py holds the two against each other)
        self._native = hasattr(wb, "device_scene")
        self._obj_bytes: Dict[str, bytes] = {}
        for name, path in self.ppm_files.items():
            if self._native:
                from . import loaders
                with open(path, "rb") as f:
                    self._images[name] = loaders.uv_image_ppm(wb, f.read())
            else:
                self._images[name] = wb.uv_image(read_ppm(path))
        for ins in self.instructions:
            if "define" in ins and "add" not in ins:
                self._define(ins)
                continue
            kind = ins.get("add")
            if kind == "camera":
                continue
            if kind == "point-light":
                wb.add_point_light(ins["at"], ins["intensity"])
            elif kind == "area-light":
                wb.add_area_light(ins["corner"], ins["uvec"], int(ins["usteps"]), ins["vvec"], int(ins["vsteps"]),
                                  ins["intensity"])
            elif kind == "group":
                g = self._make_group(ins)
                wb.add_group(g, cache_bounds=True)  # scene.rs:195-198
            elif kind == "csg":
                wb.add_csg(self._make_csg(ins))
            elif kind in B.SHAPE_KINDS:
                wb.add_shape(self._make_shape(ins, kind))
            else:
                raise ValueError(f"unknown instruction: {ins!r}")
        return wb

    # -- define / extend (scene.rs:152-181) -------------------------------------------
    def _define(self, ins):
        name, extend = ins["define"], ins.get("extend")
        specs = parse_transform(ins.get("transform"))
        if specs:  # deser_transform maps an empty list to None
            final = []
            for d in extend or []:
                final.extend(self._define_transforms.get(d, []))
            final.extend(specs)
            self._define_transforms[name] = final
        if ins.get("material") is not None:
            final = []
            for d in extend or []:
                final.extend(self._define_materials.get(d, []))
            final.append(ins["material"])
            self._define_materials[name] = final

    # -- make_transform_material (scene.rs:629-661) ----------------------------------
    def _transform_material(self, ins):
        ops: list = []
        mat = B.Material()
        for d in ins.get("extend") or []:
            ops.extend(self._define_transforms.get(d, []))
            for spec in self._define_materials.get(d, []):
                mat = self._update_material(mat, spec)
        ops.extend(parse_transform(ins.get("transform")))
        if ins.get("material") is not None:
            mat = self._update_material(mat, ins["material"])
        return ops, mat

    def _update_material(self, mat: B.Material, spec: dict) -> B.Material:  # scene.rs:1159-1190
        m = B.Material(**vars(mat))
        if spec.get("pattern") is not None:
            m.pattern = self._make_pattern(spec["pattern"])
        for key in _MATERIAL_FIELDS:
            if spec.get(key) is not None:
                setattr(m, key.replace("-", "_"), spec[key])
        return m

    def _make_uv(self, spec: dict) -> int:  # scene.rs:1325-1348
        kind = spec["kind"]
        if kind == "checker":
            return self._wb.uv_checker(float(spec["width"]), float(spec["height"]), spec["colors"][0],
                                       spec["colors"][1])
        if kind == "align-check":
            return self._wb.uv_align_check(spec["main"], spec["ul"], spec["ur"], spec["bl"], spec["br"])
        if kind == "image":
            if spec["ppm"] not in self._images:
                raise KeyError(f"Couldn't find ppm file named: {spec['ppm']}")
            return self._images[spec["ppm"]]
        raise ValueError(f"unknown uv pattern kind {kind!r}")

    def _make_pattern(self, spec: dict) -> int:  # scene.rs:1241-1302
        kind = _PATTERN_KINDS[spec["kind"]]
        ops = parse_transform(spec.get("transform"))
        if kind == B.PAT_TEXTURE_MAP:
            return self._wb.pattern(kind, None, ops, _UV_MAPPINGS[spec["uv-mapping"]],
                                    [self._make_uv(spec["uv-pattern"])])
        if kind == B.PAT_CUBE_MAP:
            uvs = [self._make_uv(spec[f]) for f in ("left", "front", "right", "back", "up", "down")]
            return self._wb.pattern(kind, None, ops, 0, uvs)
        if kind == B.PAT_XYZ_RGB:
            return self._wb.pattern(kind, None, ops)
        return self._wb.pattern(kind, spec["colors"], ops)

    # -- shapes (scene.rs:384-568) ------------------------------------------------------
    def _make_shape(self, ins: dict, kind: str) -> int:
        ops, mat = self._transform_material(ins)
        shadow = ins.get("shadow")
        shadow = True if shadow is None else bool(shadow)
        params = None
        if kind in ("cylinder", "cone"):
            mn = ins.get("min")
            mx = ins.get("max")
            params = [-math.inf if mn is None else mn, math.inf if mx is None else mx,
                      1.0 if ins.get("closed") else 0.0]
        elif kind == "triangle":
            params = [*ins["p1"], *ins["p2"], *ins["p3"]]
        elif kind == "smooth-triangle":
            params = [*ins["p1"], *ins["p2"], *ins["p3"], *ins["n1"], *ins["n2"], *ins["n3"]]
        return self._wb.shape(kind, ops, mat, shadow, params)

    # -- groups (scene.rs:570-626) -------------------------------------------------------
    def _mesh(self, name: str) -> list:
        if name in self.meshes:
            return self.meshes[name]
        if name not in self._mesh_cache:
            if name not in self.obj_files:
                raise FileNotFoundError(f"Couldn't find file named {name}")
            self._mesh_cache[name] = parse_obj(self.obj_files[name])
        return self._mesh_cache[name]

    def _make_group(self, ins: dict) -> int:
        wb = self._wb
        ops, mat = self._transform_material(ins)
        g = wb.group(())
        if ins.get("obj") is not None and self._native and ins["obj"] not in self.meshes:
            from . import loaders
            name = ins["obj"]
            if name not in self._obj_bytes:
                if name not in self.obj_files:
                    raise FileNotFoundError(f"Couldn't find file named {name}")
                with open(self.obj_files[name], "rb") as f:
                    self._obj_bytes[name] = f.read()
            loaders.group_add_obj(wb, g, mat, self._obj_bytes[name])  # parse_obj (obj.rs:8-41), natively
        elif ins.get("obj") is not None:
            for og in self._mesh(ins["obj"]):  # parse_obj (obj.rs:8-41): one child group per OBJ group
                wb.group_add_mesh(g, mat, og["positions"], og["normals"], og["vidx"], og["nidx"])
        wb.group_set_transform(g, ops)  # `.with_transform(transform)` (scene.rs:602)
        if ins.get("divide") is not None:
            wb.group_divide(g, int(ins["divide"]))  # before explicit shapes (scene.rs:604-620, Q8)
        for child in ins.get("shapes") or []:
            if child.get("add") == "group":
                wb.group_add_child(g, self._make_group(child))
            else:
                wb.group_add_shape(g, self._make_shape(child, child["add"]))
        return g

    # -- csg (scene.rs:663-786) ---------------------------------------------------------
    def _make_csg(self, ins: dict) -> int:
        def child(elem):
            if elem["kind"] == "csg":
                return B.CHILD_CSG, self._make_csg(elem)
            return B.CHILD_SHAPE, self._make_shape(elem, elem["kind"])

        lk, l = child(ins["args"][0])
        rk, r = child(ins["args"][1])
        return self._wb.csg(_CSG_OPS[ins["op"]], lk, l, rk, r)
