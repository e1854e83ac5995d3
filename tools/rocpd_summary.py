#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) output into the text kept under profiles/.

usage: rocpd_summary.py <results.db> [...]  > profiles/<name>.txt
For every db: per-kernel call count / total / average / min / max duration (the same numbers as
`rocprofv3 --stats`), register and scratch use of each kernel, and, when the run collected PMC
counters, the per-dispatch counter values of the render kernels.
"""
import sqlite3
import sys


def short(name):
    """The kernel's own name with its template arguments: no return type, namespace or parameter list."""
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return name if len(name) <= 60 else name[:57] + "..."


def summarise(path: str) -> None:
    db = sqlite3.connect(path)
    print(f"== {path}")
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(grid_y), "
        "max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<60} {'calls':>5} {'total_ms':>10} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'%':>6} "
          f"{'avgpr*':>6} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>6} grid(threads) wg")
    for r in rows:
        name = short(r[0])
        print(f"{name:<60} {r[1]:>5} {r[2] / 1e6:>10.3f} {r[3] / 1e6:>10.3f} {r[4] / 1e6:>10.3f} {r[5] / 1e6:>10.3f} "
              f"{100.0 * r[2] / total:>6.2f} {r[6]:>6} {r[7]:>5} {r[8]:>5} {r[9]:>8} {r[10]:>6} {r[11]}x{r[12]} {r[13]}")
    print("  (* rocpd's arch_vgpr_count, NOT the allocation: the code object's .vgpr_count / spills are listed by "
          "tools/kernel_resources.py)")
    pmc = db.execute(
        "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name order by avg(duration) desc").fetchall()
    if pmc:
        print("-- PMC counters (per dispatch)")
        print(f"{'kernel':<60} {'counter':<20} {'n':>4} {'avg':>18} {'min':>18} {'max':>18} {'avg_ms':>10}")
        for r in pmc:
            name = short(r[0])
            print(f"{name:<60} {r[1]:<20} {r[2]:>4} {r[3]:>18.1f} {r[4]:>18.1f} {r[5]:>18.1f} {r[6] / 1e6:>10.3f}")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
