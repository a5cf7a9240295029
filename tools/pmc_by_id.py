#!/usr/bin/env python3
"""Attribute the rocprofv3 passes over tools/pmc_replay.py to the train step's launch ids.

    python tools/pmc_by_id.py <dir with pass sub-directories> <out.json> --ids 15,8,... --iters 5 [--trace-iters 21]

pmc_replay.py issues, after its set-up, `iters` replays of each id in order and then 3 operand splits + `iters` NT GEMMs;
the library's dispatches (everything that is not a torch / runtime kernel) are consumed from the END of each pass in that
order, and a block whose dispatches do not all carry one kernel name stops the run.  Corrections as tools/pmc_collect.py
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 on gfx950, both counters in KB, the first launch of a block dropped."""
import argparse
import csv
import glob
import json
import re
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def is_library(name):
    return not any(s in name for s in ("at::", "rocclr", "rocprim", "hipcub", "__amd_"))


def blocks(rows, ids, iters):
    """rows: [(dispatch_id, kernel, {counter: value})] in dispatch order -> {id: [rows of its block]}"""
    lib = [r for r in rows if is_library(r[1])]
    need = iters * len(ids) + 3 + iters
    assert len(lib) >= need, (len(lib), need)
    tail = lib[-need:]
    out, pos = {}, 0
    for lid in ids:
        blk = tail[pos:pos + iters]
        pos += iters
        names = {b[1] for b in blk}
        assert len(names) == 1, ("launch id", lid, "does not map to one kernel", names)
        out[lid] = blk
    gemm = tail[-iters:]
    assert len({b[1] for b in gemm}) == 1 and "gemm" in gemm[0][1], gemm[0][1]
    out["gemm"] = gemm
    return out


def read_pass(path, trace):
    per = {}
    for r in csv.DictReader(open(path)):
        if trace:
            k = int(r["Dispatch_Id"])
            per[k] = (k, short(r["Kernel_Name"]), {"duration_ns": float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))})
        else:
            k = int(r["Dispatch_Id"])
            e = per.setdefault(k, (k, short(r["Kernel_Name"]), {}))
            e[2][r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("out")
    ap.add_argument("--ids", required=True)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--trace-iters", type=int, default=21)
    ap.add_argument("--names", default=None, help="JSON {id: launch name} written by pmc_replay.py --names-out")
    ap.add_argument("--script", default="tools/roundend5.sh", help="the evidence script that ran the passes (recorded in the output)")
    args = ap.parse_args()
    ids = [int(v) for v in args.ids.split(",")]
    agg = defaultdict(dict)
    kern = {}
    for f in sorted(glob.glob(args.root + "/**/*counter_collection.csv", recursive=True)):
        for lid, blk in blocks(read_pass(f, False), ids, args.iters).items():
            kern[lid] = blk[0][1]
            use = blk[1:] if len(blk) > 2 else blk
            for c in use[0][2]:
                agg[lid][c] = sum(b[2][c] for b in use) / len(use)
    for f in sorted(glob.glob(args.root + "/**/*kernel_trace.csv", recursive=True)):
        for lid, blk in blocks(read_pass(f, True), ids, args.trace_iters).items():
            use = blk[1:]
            agg[lid]["duration_ns"] = sum(b[2]["duration_ns"] for b in use) / len(use)
    names = json.load(open(args.names)) if args.names else {}
    res = {}
    for lid, e in agg.items():
        e = dict(e)
        e["kernel"] = kern.get(lid, "")
        if str(lid) in names:
            e["launch"] = names[str(lid)]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
        if "SQ_INSTS_VALU" in e and "SQ_INSTS_MFMA" in e:
            e["valu_per_mfma"] = round(e["SQ_INSTS_VALU"] / max(1.0, e["SQ_INSTS_MFMA"]), 2)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CU_CYCLES" in e:
            e["mfma_busy_frac_of_cu_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, 4.0 * e["SQ_BUSY_CU_CYCLES"]), 4)
        if "duration_ns" in e and "hbm_bytes_per_launch" in e:
            e["hbm_gbs"] = round(e["hbm_bytes_per_launch"] / e["duration_ns"], 1)
        res[str(lid)] = e
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tinydiffusionmodels_amd.build import source_digest
    doc = {"command": f"{args.script}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) and --kernel-trace, "
                      "each directly in front of python tools/pmc_replay.py (in-pipeline arguments, B=512, two alternating workspaces)",
           "source_digest": source_digest(),      # bench.py carries `traffic` only when its own library has this digest
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), KB -> bytes; first launch of each block dropped",
           "by_launch_id": {k: v for k, v in res.items() if k != "gemm"}, "text_ffn1_gemm": res.get("gemm")}
    json.dump(doc, open(args.out, "w"), indent=1)
    for k, e in res.items():
        print(k, e["kernel"][:44], {c: (round(v, 1) if isinstance(v, float) else v) for c, v in e.items()
                                    if c in ("duration_ns", "hbm_bytes_per_launch", "l2_hit_rate", "valu_per_mfma", "mfma_busy_frac_of_cu_busy", "hbm_gbs")})


if __name__ == "__main__":
    main()
