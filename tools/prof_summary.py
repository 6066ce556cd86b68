#!/usr/bin/env python3
"""Summarise rocprofv3 output databases (kernel trace + PMC passes) into a small text file for profiles/.

    python tools/prof_summary.py gpurun_out/prof_trace/bench_results.db gpurun_out/prof_pmc_*/pmc_results.db > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main(paths):
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        tabs = {r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")}
        print(f"== {p}")
        if "top_kernels" in tabs:
            print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
            for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                print(f"{name[:70]:70s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
        if "counters_collection" in tabs:
            q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), max(vgpr_count), "
                 "max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), max(workgroup_size) "
                 "from counters_collection where kernel_name like '%wekws%' group by kernel_name, counter_name")
            rows = list(cur.execute(q))
            seen = set()
            for r in rows:
                if r[0] not in seen:
                    seen.add(r[0])
                    print(f"kernel {r[0][:90]}  vgpr={r[6]} agpr={r[7]} sgpr={r[8]} lds={r[9]} grid={r[10]} wg={r[11]}")
            print(f"{'counter':28s} {'n':>4s} {'avg':>16s} {'min':>16s} {'max':>16s}")
            for r in rows:
                print(f"{r[1]:28s} {r[2]:4d} {r[3]:16.1f} {r[4]:16.1f} {r[5]:16.1f}   [{r[0][:48]}]")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
