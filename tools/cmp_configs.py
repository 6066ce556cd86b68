#!/usr/bin/env python3
"""Side-by-side of two tools/bench_configs.py outputs:  python tools/cmp_configs.py old.jsonl new.jsonl"""
import json
import sys


def load(path):
    rows = {}
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        key = tuple(str(r.get(k, "")) for k in ("kind", "model", "pipeline", "mode", "B", "T", "chunk", "lds_cache", "nsamp"))
        rows[key] = r
    return rows


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    for key, ra in a.items():
        rb = b.get(key)
        f = "ms_per_chunk_stream" if "ms_per_chunk_stream" in ra else "ms"
        if rb is None or f not in ra:
            continue
        name = " ".join(k for k in key if k)
        print(f"{name:70s} {ra[f]:9.4f} -> {rb[f]:9.4f} ms  {100.0 * (rb[f] / ra[f] - 1.0):+6.1f} %")


if __name__ == "__main__":
    main()
