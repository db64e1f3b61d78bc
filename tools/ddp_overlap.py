#!/usr/bin/env python3
"""Overlap of the RCCL all-reduce kernels with the backward's kernels, from a rocprofv3 --kernel-trace CSV of the
bench (MMTG_FORCE_DDP=1 on one GPU, or a real multi-GPU run).

    python tools/ddp_overlap.py <..._kernel_trace.csv>

For every RCCL kernel dispatch: its duration and the share of it during which at least one non-RCCL kernel of the same
process was executing (interval intersection on the trace's start / end timestamps)."""
import csv
import json
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    ks = []
    for r in rows:
        name = r.get("Kernel_Name", "")
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    ks.sort()
    is_rccl = lambda n: "ccl" in n.lower() or "AllReduce" in n or "ncclDevKernel" in n
    comp = [(s, e) for s, e, n, q, st in ks if not is_rccl(n)]
    rccl = [(s, e, n, q) for s, e, n, q, st in ks if is_rccl(n)]
    # merged compute intervals
    merged = []
    for s, e in comp:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [m[0] for m in merged]
    out = []
    for s, e, n, q in rccl:
        i = max(0, bisect.bisect_right(starts, s) - 1)
        ov = 0
        while i < len(merged) and merged[i][0] < e:
            ov += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        out.append((e - s, ov))
    tot = sum(d for d, _ in out)
    ovl = sum(o for _, o in out)
    queues = sorted(set(q for _, _, _, q in rccl))
    cq = sorted(set(q for s, e, n, q, st in ks if not is_rccl(n)))
    res = {"rccl_kernel_dispatches": len(out), "rccl_kernel_time_ms": round(tot / 1e6, 3),
           "rccl_time_overlapped_by_engine_kernels_ms": round(ovl / 1e6, 3),
           "overlap_fraction": round(ovl / tot, 4) if tot else None,
           "rccl_kernel_names": sorted(set(n[:80] for _, _, n, _ in rccl))[:6],
           "rccl_queues": queues, "engine_queues": cq,
           "engine_kernel_dispatches": len(comp), "trace": path}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
