// rt_loaders.cpp — include/rt_loaders.h: OBJ in, PPM in, PPM out, native and threaded.
//
// parse_obj (src/obj.rs:8-41) goes through the third-party crate `wavefront` 0.2, Canvas::from_ppm / to_ppm
// (src/canvas.rs:57-107) through `image` 0.24; neither crate is in the reference tree.  What is restated here is what
// the reference's own call sites and tests pin: fan triangulation (obj.rs:82-123), groups by name (obj.rs:125-158),
// normals per corner (obj.rs:160-205), sample / maxval (canvas.rs:158-170), clamp-scale-round quantisation
// (canvas.rs:73-75, done on the device by rt_render_u8 before the bytes get here).
//
// Part of librt_host.so; calls the world-builder entry points of rt_builder.h like any other host would.
#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#define RTB_PREFIX(name) rth_##name
#include "../../../include/rt_loaders.h"

namespace {

thread_local std::string g_error;
int fail(const std::string& msg, int code = -1) {
  g_error = msg;
  return code;
}
// asked > 0: that many threads (as long as each gets something); otherwise one per hardware thread, at most 16, and no
// more than the work is worth
int thread_count(int asked, size_t work_items, size_t min_per_thread) {
  if (asked > 0) return (int)std::max<size_t>(1, std::min<size_t>((size_t)asked, work_items / 64 + 1));
  const int n = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  const size_t by_work = std::max<size_t>(1, work_items / std::max<size_t>(1, min_per_thread));
  return (int)std::min<size_t>((size_t)n, by_work);
}

// ---------------------------------------------------------------------------------------------------------------------
// OBJ
// ---------------------------------------------------------------------------------------------------------------------
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r'; }
inline const char* skip_space(const char* p, const char* e) {
  while (p < e && is_space(*p)) p++;
  return p;
}
inline const char* token_end(const char* p, const char* e) {
  while (p < e && !is_space(*p)) p++;
  return p;
}
// Decimal text to the nearest f32 (what Rust's str::parse::<f32> gives); accepts a leading '+', which from_chars does not.
inline bool parse_f32(const char* p, const char* e, float* out) {
  if (p < e && *p == '+') p++;
  auto r = std::from_chars(p, e, *out);
  return r.ec == std::errc() && r.ptr == e;
}
inline bool parse_i64(const char* p, const char* e, long long* out) {
  if (p < e && *p == '+') p++;
  auto r = std::from_chars(p, e, *out);
  return r.ec == std::errc() && r.ptr == e;
}

struct Corner {
  long long v, n;  // raw OBJ indices (1-based, negative = relative), n = 0: none
};
// What one chunk of the file holds, in file order: faces are cut into runs by the `g` lines between them.
struct FaceRun {
  bool inherits;     // no `g` line in this chunk before these faces: they belong to the group open at the chunk's start
  std::string name;  // otherwise the group's name
  std::vector<int32_t> vidx, nidx;  // resolved 0-based indices, 3 per triangle
};
struct Chunk {
  const char* begin;
  const char* end;
  size_t n_v = 0, n_vn = 0;      // pass 1
  size_t base_v = 0, base_vn = 0;  // positions / normals before this chunk
  std::vector<FaceRun> runs;
  std::string error;
};

template <class F>
void for_each_line(const char* p, const char* e, F&& f) {
  while (p < e) {
    const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(e - p)));
    const char* le = nl ? nl : e;
    f(p, le);
    p = nl ? nl + 1 : e;
  }
}

void count_chunk(Chunk& c) {
  for_each_line(c.begin, c.end, [&](const char* p, const char* e) {
    p = skip_space(p, e);
    if (e - p >= 2 && p[0] == 'v') {
      if (is_space(p[1]))
        c.n_v++;
      else if (p[1] == 'n' && e - p >= 3 && is_space(p[2]))
        c.n_vn++;
    }
  });
}

void parse_chunk(Chunk& c, float* positions, float* normals) {
  size_t nv = c.base_v, nvn = c.base_vn;
  FaceRun* run = nullptr;
  std::vector<Corner> corners;
  size_t line_no = 0;
  auto bad = [&](const char* what) {
    if (c.error.empty()) c.error = std::string(what) + " (line " + std::to_string(line_no) + " of its chunk)";
  };
  for_each_line(c.begin, c.end, [&](const char* p, const char* e) {
    line_no++;
    p = skip_space(p, e);
    if (p == e || *p == '#') return;
    const char* te = token_end(p, e);
    const std::string_view tag(p, (size_t)(te - p));
    if (tag == "v" || tag == "vn") {
      float xyz[3];
      const char* q = te;
      for (int k = 0; k < 3; k++) {
        q = skip_space(q, e);
        const char* qe = token_end(q, e);
        if (q == qe || !parse_f32(q, qe, &xyz[k])) return bad("malformed vertex");
        q = qe;
      }
      float* dst = tag == "v" ? positions + 3 * nv++ : normals + 3 * nvn++;
      dst[0] = xyz[0], dst[1] = xyz[1], dst[2] = xyz[2];
    } else if (tag == "g") {
      const char* q = skip_space(te, e);
      const char* qe = token_end(q, e);
      c.runs.push_back(FaceRun{false, std::string(q, (size_t)(qe - q)), {}, {}});
      run = &c.runs.back();
    } else if (tag == "f") {
      corners.clear();
      const char* q = te;
      while (true) {
        q = skip_space(q, e);
        if (q == e) break;
        const char* qe = token_end(q, e);
        // v, v/vt, v//vn, v/vt/vn
        const char* s1 = static_cast<const char*>(std::memchr(q, '/', (size_t)(qe - q)));
        Corner cr{0, 0};
        if (!parse_i64(q, s1 ? s1 : qe, &cr.v) || cr.v == 0) return bad("malformed face index");
        if (s1) {
          const char* s2 = static_cast<const char*>(std::memchr(s1 + 1, '/', (size_t)(qe - s1 - 1)));
          if (s2 && s2 + 1 < qe && (!parse_i64(s2 + 1, qe, &cr.n) || cr.n == 0)) return bad("malformed face normal index");
        }
        corners.push_back(cr);
        q = qe;
      }
      if (corners.size() < 3) return bad("face with fewer than three corners");
      if (!run) {
        c.runs.push_back(FaceRun{true, std::string(), {}, {}});
        run = &c.runs.back();
      }
      // 1-based or relative to the tables as they stand at this line
      auto resolve = [](long long i, size_t count) { return i > 0 ? i - 1 : (long long)count + i; };
      int32_t v[16], n[16];
      std::vector<int32_t> vv, nn;
      int32_t* pv = v;
      int32_t* pn = n;
      if (corners.size() > 16) {
        vv.resize(corners.size());
        nn.resize(corners.size());
        pv = vv.data();
        pn = nn.data();
      }
      for (size_t k = 0; k < corners.size(); k++) {
        const long long vi = resolve(corners[k].v, nv);
        if (vi < 0 || (size_t)vi >= nv) return bad("face refers to a vertex that is not defined yet");
        pv[k] = (int32_t)vi;
        pn[k] = -1;
        if (corners[k].n != 0) {
          const long long ni = resolve(corners[k].n, nvn);
          if (ni < 0 || (size_t)ni >= nvn) return bad("face refers to a normal that is not defined yet");
          pn[k] = (int32_t)ni;
        }
      }
      for (size_t k = 1; k + 1 < corners.size(); k++) {  // fan around the first corner (obj.rs:82-123)
        run->vidx.insert(run->vidx.end(), {pv[0], pv[k], pv[k + 1]});
        run->nidx.insert(run->nidx.end(), {pn[0], pn[k], pn[k + 1]});
      }
    }
    // o, s, usemtl, mtllib, vt, ...: not read by parse_obj
  });
}

struct ObjData {
  std::vector<float> positions, normals;
  std::vector<int32_t> vidx, nidx;
  std::vector<uint32_t> group_first;
  std::vector<std::string> names;
  std::vector<const char*> name_ptrs;
};

int parse_obj(const char* text, size_t len, int threads, ObjData& out) {
  const int nt = thread_count(threads, len, 1u << 20);
  std::vector<Chunk> chunks((size_t)nt);
  {
    const char* p = text;
    const char* e = text + len;
    for (int k = 0; k < nt; k++) {
      const char* want = k + 1 == nt ? e : text + len * (size_t)(k + 1) / (size_t)nt;
      if (want < p) want = p;
      if (want < e) {  // cut at a line end
        const char* nl = static_cast<const char*>(std::memchr(want, '\n', (size_t)(e - want)));
        want = nl ? nl + 1 : e;
      }
      chunks[k].begin = p;
      chunks[k].end = want;
      p = want;
    }
  }
  auto run_all = [&](auto&& fn) {
    if (nt == 1) {
      fn(chunks[0]);
      return;
    }
    std::vector<std::thread> pool;
    for (int k = 0; k < nt; k++) pool.emplace_back([&, k] { fn(chunks[k]); });
    for (auto& t : pool) t.join();
  };
  run_all([](Chunk& c) { count_chunk(c); });
  size_t nv = 0, nvn = 0;
  for (Chunk& c : chunks) {
    c.base_v = nv;
    c.base_vn = nvn;
    nv += c.n_v;
    nvn += c.n_vn;
  }
  if (nv > 0x7FFFFFFFu / 3 || nvn > 0x7FFFFFFFu / 3) return fail("OBJ file too large");
  out.positions.assign(3 * nv, 0.f);
  out.normals.assign(3 * nvn, 0.f);
  run_all([&](Chunk& c) { parse_chunk(c, out.positions.data(), out.normals.data()); });
  for (size_t k = 0; k < chunks.size(); k++)
    if (!chunks[k].error.empty()) return fail("OBJ: " + chunks[k].error + ", chunk " + std::to_string(k));
  // groups by name, in order of first appearance; a run that inherits belongs to the group open at its chunk's start
  std::unordered_map<std::string, size_t> index;
  std::vector<std::vector<const FaceRun*>> members;
  std::string current;  // faces before any `g` line: the default group
  for (const Chunk& c : chunks)
    for (const FaceRun& r : c.runs) {
      if (!r.inherits) current = r.name;
      if (r.vidx.empty()) continue;  // a `g` line without faces opens nothing (groups exist through their faces)
      auto it = index.find(current);
      if (it == index.end()) {
        it = index.emplace(current, members.size()).first;
        members.emplace_back();
        out.names.push_back(current);
      }
      members[it->second].push_back(&r);
    }
  size_t total = 0;
  for (auto& m : members)
    for (const FaceRun* r : m) total += r->vidx.size();
  if (total / 3 > 0x7FFFFFFFu) return fail("OBJ file too large");
  out.vidx.reserve(total);
  out.nidx.reserve(total);
  out.group_first.push_back(0);
  for (auto& m : members) {
    for (const FaceRun* r : m) {
      out.vidx.insert(out.vidx.end(), r->vidx.begin(), r->vidx.end());
      out.nidx.insert(out.nidx.end(), r->nidx.begin(), r->nidx.end());
    }
    out.group_first.push_back((uint32_t)(out.vidx.size() / 3));
  }
  for (auto& s : out.names) out.name_ptrs.push_back(s.c_str());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// PPM
// ---------------------------------------------------------------------------------------------------------------------
// Next header / sample token of a PNM stream: skips whitespace and `# ...` comments.
bool ppm_token(const uint8_t*& p, const uint8_t* e, const uint8_t** tb, const uint8_t** te) {
  while (p < e) {
    if (*p == '#') {
      while (p < e && *p != '\n') p++;
    } else if (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') {
      p++;
    } else {
      break;
    }
  }
  if (p >= e) return false;
  *tb = p;
  while (p < e && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '#')) p++;
  *te = p;
  return true;
}
bool ppm_uint(const uint8_t*& p, const uint8_t* e, uint32_t* v) {
  const uint8_t *tb, *te;
  if (!ppm_token(p, e, &tb, &te)) return false;
  auto r = std::from_chars(reinterpret_cast<const char*>(tb), reinterpret_cast<const char*>(te), *v);
  return r.ec == std::errc() && r.ptr == reinterpret_cast<const char*>(te);
}

int ppm_decode(const uint8_t* data, size_t len, uint32_t* w, uint32_t* h, std::vector<float>& rgb) {
  const uint8_t* p = data;
  const uint8_t* e = data + len;
  const uint8_t *tb, *te;
  if (!ppm_token(p, e, &tb, &te) || te - tb != 2 || tb[0] != 'P' || (tb[1] != '3' && tb[1] != '6'))
    return fail("PPM: only P3 and P6 are read (Canvas::from_ppm decodes to RGB)");
  const bool ascii = tb[1] == '3';
  uint32_t maxval = 0;
  if (!ppm_uint(p, e, w) || !ppm_uint(p, e, h) || !ppm_uint(p, e, &maxval)) return fail("PPM: malformed header");
  if (*w == 0 || *h == 0 || maxval == 0) return fail("PPM: empty image or maxval 0");
  if (maxval > 255)
    return fail("PPM: maxval > 255: what the reference's decoder does with 16-bit samples before dividing is not pinned");
  // (w * h * 3 floats must be addressable: a header that promises more is refused before anything is sized by it)
  if ((uint64_t)*w * (uint64_t)*h > (uint64_t)SIZE_MAX / 12u) return fail("PPM: image too large");
  const size_t n = (size_t)*w * *h * 3;
  if (!ascii) {
    // P6: exactly one whitespace byte separates maxval from the samples; a file that ends at the maxval token, or in
    // the middle of the samples, is truncated (checked before the buffer is sized or read)
    if (p >= e || !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f'))
      return fail("PPM: truncated (no whitespace byte after maxval)");
    if ((size_t)(e - (p + 1)) < n) return fail("PPM: fewer samples than the header promises");
  }
  rgb.resize(n);
  const float mv = (float)maxval;
  if (ascii) {
    // the samples, on threads: the body is cut at line ends (a comment never crosses one), every piece counts its
    // samples, then parses them to where they belong
    const int nt = thread_count(0, (size_t)(e - p), 1u << 20);
    struct Piece {
      const uint8_t *b, *e;
      size_t count = 0, first = 0;
      int error = 0;  // 1: not a number, 2: above maxval
    };
    std::vector<Piece> pieces((size_t)nt);
    {
      const uint8_t* q = p;
      for (int k = 0; k < nt; k++) {
        const uint8_t* want = k + 1 == nt ? e : p + (size_t)(e - p) * (size_t)(k + 1) / (size_t)nt;
        if (want < q) want = q;
        if (want < e) {
          const uint8_t* nl = static_cast<const uint8_t*>(std::memchr(want, '\n', (size_t)(e - want)));
          want = nl ? nl + 1 : e;
        }
        pieces[k].b = q;
        pieces[k].e = want;
        q = want;
      }
    }
    auto on_threads = [&](auto&& fn) {
      if (nt == 1) return fn(pieces[0]);
      std::vector<std::thread> pool;
      for (int k = 0; k < nt; k++) pool.emplace_back([&, k] { fn(pieces[k]); });
      for (auto& t : pool) t.join();
    };
    on_threads([](Piece& pc) {
      const uint8_t* q = pc.b;
      const uint8_t *tb, *te;
      while (ppm_token(q, pc.e, &tb, &te)) pc.count++;
    });
    size_t total = 0;
    for (Piece& pc : pieces) {
      pc.first = total;
      total += pc.count;
    }
    if (total < n) return fail("PPM: fewer samples than the header promises");
    float* out = rgb.data();
    on_threads([&](Piece& pc) {
      const uint8_t* q = pc.b;
      for (size_t i = pc.first; i < pc.first + pc.count && i < n; i++) {
        uint32_t v;
        if (!ppm_uint(q, pc.e, &v)) {
          pc.error = 1;
          return;
        }
        if (v > maxval) {
          pc.error = 2;
          return;
        }
        out[i] = (float)v / mv;  // canvas.rs:99-101: u8 sample as f32 / maxval as f32
      }
    });
    for (const Piece& pc : pieces) {
      if (pc.error == 1) return fail("PPM: malformed sample");
      if (pc.error == 2) return fail("PPM: sample above maxval");
    }
  } else {
    p++;  // the single whitespace byte after maxval (present, and n bytes behind it: checked above)
    for (size_t i = 0; i < n; i++) rgb[i] = (float)p[i] / mv;
  }
  return 0;
}

// one row of samples: "r g b r g b ...\n"
inline char* put_u8(char* o, uint8_t v) {
  if (v >= 100) {
    *o++ = (char)('0' + v / 100);
    v %= 100;
    *o++ = (char)('0' + v / 10);
    *o++ = (char)('0' + v % 10);
  } else if (v >= 10) {
    *o++ = (char)('0' + v / 10);
    *o++ = (char)('0' + v % 10);
  } else {
    *o++ = (char)('0' + v);
  }
  return o;
}
size_t format_rows(const uint8_t* rgb, uint32_t width, uint32_t r0, uint32_t r1, char* out) {
  char* o = out;
  const size_t per_row = (size_t)width * 3;
  for (uint32_t r = r0; r < r1; r++) {
    const uint8_t* s = rgb + (size_t)r * per_row;
    for (size_t k = 0; k < per_row; k++) {
      o = put_u8(o, s[k]);
      *o++ = k + 1 == per_row ? '\n' : ' ';
    }
  }
  return (size_t)(o - out);
}

// Formats the frame in bands on threads; `sink` receives the header, then every band in order.
template <class Sink>
int ppm_encode(const uint8_t* rgb, uint32_t width, uint32_t height, int threads, Sink&& sink) {
  if (!rgb || width == 0 || height == 0) return fail("PPM: empty image");
  char header[64];
  const int hl = std::snprintf(header, sizeof(header), "P3\n%u %u\n255\n", width, height);
  const int nt = thread_count(threads, (size_t)width * height, 1u << 16);
  std::vector<std::vector<char>> bands((size_t)nt);
  std::vector<size_t> used((size_t)nt, 0);
  auto work = [&](int k) {
    const uint32_t r0 = (uint32_t)((uint64_t)height * k / nt), r1 = (uint32_t)((uint64_t)height * (k + 1) / nt);
    bands[k].resize((size_t)(r1 - r0) * width * 12 + 1);  // at most "255 " per sample
    used[k] = format_rows(rgb, width, r0, r1, bands[k].data());
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int k = 0; k < nt; k++) pool.emplace_back(work, k);
    for (auto& t : pool) t.join();
  }
  if (!sink(header, (size_t)hl)) return -1;
  for (int k = 0; k < nt; k++)
    if (!sink(bands[k].data(), used[k])) return -1;
  return 0;
}

}  // namespace

extern "C" {

const char* rtl_last_error(void) { return g_error.c_str(); }

int rtl_obj_parse(const char* text, size_t len, int threads, RtlObj* out) {
  if (!out || (!text && len)) return fail("null argument");
  ObjData* d = new ObjData();
  const int rc = parse_obj(text, len, threads, *d);
  if (rc != 0) {
    delete d;
    return rc;
  }
  out->n_positions = (uint32_t)(d->positions.size() / 3);
  out->n_normals = (uint32_t)(d->normals.size() / 3);
  out->n_triangles = (uint32_t)(d->vidx.size() / 3);
  out->n_groups = (uint32_t)d->names.size();
  out->positions = d->positions.data();
  out->normals = d->normals.data();
  out->vidx = d->vidx.data();
  out->nidx = d->nidx.data();
  out->group_first = d->group_first.data();
  out->group_names = d->name_ptrs.data();
  out->owner = d;
  return 0;
}

void rtl_obj_free(RtlObj* o) {
  if (!o) return;
  delete static_cast<ObjData*>(o->owner);
  std::memset(o, 0, sizeof(*o));
}

int rtl_group_add_obj(RtbWorld* world, int group, const RtbMaterial* material, const char* text, size_t len, int threads) {
  if (!world) return fail("null world");
  ObjData d;
  const int rc = parse_obj(text, len, threads, d);
  if (rc != 0) return rc;
  const bool has_n = !d.normals.empty();
  // A file with one group hands its tables over as they are.  With several groups each gets tables of the vertices it
  // uses (in order of first use, indices remapped): rth_group_add_mesh hashes and keeps the tables it is given per mesh,
  // so whole-file tables would cost O(groups x vertices) time and memory (CAD exports have thousands of `g` sections).
  const bool compact = d.names.size() > 1;
  std::vector<int32_t> pos_map, nrm_map;  // file index -> index in the group's table, stamped per group
  std::vector<uint32_t> pos_stamp, nrm_stamp;
  if (compact) {
    pos_map.resize(d.positions.size() / 3);
    pos_stamp.assign(d.positions.size() / 3, 0xFFFFFFFFu);
    nrm_map.resize(d.normals.size() / 3);
    nrm_stamp.assign(d.normals.size() / 3, 0xFFFFFFFFu);
  }
  std::vector<float> gpos, gnrm;
  std::vector<int32_t> gv, gn;
  for (size_t g = 0; g < d.names.size(); g++) {  // one child group per OBJ group (obj.rs:12-38)
    const uint32_t t0 = d.group_first[g], t1 = d.group_first[g + 1];
    int child;
    if (!compact) {
      child = rth_group_add_mesh(world, group, material, d.positions.data(), (int)(d.positions.size() / 3),
                                 has_n ? d.normals.data() : nullptr, (int)(d.normals.size() / 3),
                                 d.vidx.data() + 3 * (size_t)t0, has_n ? d.nidx.data() + 3 * (size_t)t0 : nullptr,
                                 (int)(t1 - t0));
    } else {
      gpos.clear(), gnrm.clear(), gv.clear(), gn.clear();
      auto remap = [&](int32_t idx, std::vector<int32_t>& map, std::vector<uint32_t>& stamp, const std::vector<float>& src,
                       std::vector<float>& dst) -> int32_t {
        if (idx < 0) return idx;  // "no normal at this corner" markers pass through
        if (stamp[(size_t)idx] != (uint32_t)g) {
          stamp[(size_t)idx] = (uint32_t)g;
          map[(size_t)idx] = (int32_t)(dst.size() / 3);
          dst.insert(dst.end(), src.begin() + 3 * (size_t)idx, src.begin() + 3 * (size_t)idx + 3);
        }
        return map[(size_t)idx];
      };
      for (size_t k = 3 * (size_t)t0; k < 3 * (size_t)t1; k++) {
        gv.push_back(remap(d.vidx[k], pos_map, pos_stamp, d.positions, gpos));
        if (has_n) gn.push_back(remap(d.nidx[k], nrm_map, nrm_stamp, d.normals, gnrm));
      }
      child = rth_group_add_mesh(world, group, material, gpos.data(), (int)(gpos.size() / 3), has_n ? gnrm.data() : nullptr,
                                 (int)(gnrm.size() / 3), gv.data(), has_n ? gn.data() : nullptr, (int)(t1 - t0));
    }
    if (child < 0) return fail(std::string("rth_group_add_mesh: ") + rth_last_error(world), child);
  }
  return (int)d.names.size();
}

int rtl_ppm_decode(const uint8_t* data, size_t len, uint32_t* width, uint32_t* height, float** rgb) {
  if (!data || !width || !height || !rgb) return fail("null argument");
  std::vector<float> v;
  const int rc = ppm_decode(data, len, width, height, v);
  if (rc != 0) return rc;
  *rgb = static_cast<float*>(std::malloc(v.size() * sizeof(float)));
  if (!*rgb) return fail("out of memory");
  std::memcpy(*rgb, v.data(), v.size() * sizeof(float));
  return 0;
}

int rtl_uv_image_ppm(RtbWorld* world, const uint8_t* data, size_t len) {
  if (!world || !data) return fail("null argument");
  std::vector<float> v;
  uint32_t w = 0, h = 0;
  const int rc = ppm_decode(data, len, &w, &h, v);
  if (rc != 0) return rc;
  const int uv = rth_uv_image(world, w, h, v.data());
  if (uv < 0) return fail(std::string("rth_uv_image: ") + rth_last_error(world), uv);
  return uv;
}

int rtl_ppm_encode(const uint8_t* rgb_u8, uint32_t width, uint32_t height, int threads, char** text, size_t* len) {
  if (!text || !len) return fail("null argument");
  std::vector<char> all;
  all.reserve((size_t)width * height * 12 + 64);
  const int rc = ppm_encode(rgb_u8, width, height, threads, [&](const char* p, size_t n) {
    all.insert(all.end(), p, p + n);
    return true;
  });
  if (rc != 0) return rc;
  *text = static_cast<char*>(std::malloc(all.size() ? all.size() : 1));
  if (!*text) return fail("out of memory");
  std::memcpy(*text, all.data(), all.size());
  *len = all.size();
  return 0;
}

int rtl_ppm_write(const char* path, const uint8_t* rgb_u8, uint32_t width, uint32_t height, int threads) {
  if (!path) return fail("null path");
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return fail(std::string("cannot open ") + path);
  const int rc = ppm_encode(rgb_u8, width, height, threads, [&](const char* p, size_t n) {
    if (std::fwrite(p, 1, n, f) == n) return true;
    fail(std::string("write to ") + path + " failed");
    return false;
  });
  if (std::fclose(f) != 0 && rc == 0) return fail(std::string("write to ") + path + " failed");
  return rc;
}

void rtl_free(void* p) { std::free(p); }

}  // extern "C"
