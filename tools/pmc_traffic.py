#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes and the kernel trace of tools/pmc.sh into profiles/pmc_traffic.json --
the file bench.py reads `roofline.kernel` / `roofline.traffic` from, keyed by the sha-256 of the library file the passes
ran with (bench.py ignores the file unless it loads that very library).

    python tools/pmc_traffic.py <key>=<tag>[=<profile summary>] [...]
        key      "model/B<batch>/<precision>", e.g. ds_tcn_h256/B1024/f16x3
        tag      the <tag> given to tools/pmc.sh (reads gpurun_out/prof_<tag>_{trace,fetch,write}/)
        profile  the committed text summary of THOSE passes (profiles/rNN..._f16x3.txt): bench.py's traffic_source
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads; WRITE_SIZE is taken as reported -- tools/probe/write_calib.hip: it is exact (KiB) on
contiguous float4 fills, on the cache's 28-byte dwordx4 + dwordx3 runs and on 4-byte stores of the same rows.
"""
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def db(tag, kind):
    paths = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}", "**", "*_results.db"), recursive=True)
    if not paths:
        raise SystemExit(f"no results db for {tag}/{kind}")
    return sqlite3.connect(sorted(paths)[-1]).cursor()


def dominant_kernel(tag):
    cur = db(tag, "trace")
    rows = list(cur.execute("select name,total_calls,average from top_kernels where name like '%wekws::%' order by total_duration desc"))
    return rows[0]


def counter(tag, kind, name, kernel):
    cur = db(tag, kind)
    r = list(cur.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name=?", (name, kernel)))
    return r[0]


def main():
    args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    from wekws_amd import _capi
    sha = hashlib.sha256(open(_capi.lib_path(), "rb").read()).hexdigest()[:16]
    out = {"lib_sha16": sha,
           "formula": "(2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes; gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md, HBM)",
           "command": "tools/pmc.sh <tag> python bench.py --no-cpu-baseline --no-extras --steps 7 --warmup 2 [--precision f32]"}
    for a in args:
        key, tag, *prof = a.split("=")
        kernel, calls, avg_ns = dominant_kernel(tag)
        fetch, nf = counter(tag, "fetch", "FETCH_SIZE", kernel)
        write, nw = counter(tag, "write", "WRITE_SIZE", kernel)
        out[key] = {"kernel": kernel, "kernel_avg_ms": round(avg_ns / 1e6, 4) if avg_ns > 1e3 else round(avg_ns / 1e3, 4),
                    "calls_in_trace": calls, "FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
                    "traffic_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
                    "profile": prof[0] if prof else None}
        print(key, out[key])
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
