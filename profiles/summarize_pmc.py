#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes of profiles/collect_pmc.sh -> one small JSON (the raw per-dispatch CSVs stay in
gpurun_out/, untracked).  Derived per kernel:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the dispatch), shader cycles = GRBM_GUI_ACTIVE / 8 XCDs
  clock_ghz      = shader cycles / dispatch duration
  issued_mfma_flop = SQ_VALU_MFMA_BUSY_CYCLES x 1024: the matrix-pipe time of the kernel's MFMAs in bf16-equivalent flop (a 16x16x32
                     bf16 / f16 MFMA = 16 384 flop holds its SIMD's pipe for 16 cycles; a block-scaled 16x16x128 fp8 one 32) -- for
                     the single-shape kernels this equals SQ_INSTS_MFMA x flop per instruction (`issued_mfma_flop_by_count`)
bench.py reads this file for `roofline.mfma_busy_frac` / `ceiling_frac` / `traffic`, keyed by kernel AND library version."""
import collections
import csv
import glob
import json
import os
import sys

SIMDS, XCDS = 1024, 8
# flop per MFMA instruction of the shape each kernel family issues (16x16x32 bf16: 2*16*16*32; 32x32x16: 2*32*32*16; 16x16x4 f32: 2*16*16*4)
MFMA_FLOP = (("local_bwd", 16384), ("joint_fwd", 16384), ("conv3x3_wgrad", 32768), ("conv3x3", 16384), ("head_local_fwd_mfma", 16384),
             ("head_local_bwd_wave", 16384), ("head_local_bwd_fused", 16384))
FAMILIES = ("local_bwd_rows_kernel", "local_bwd_f8_kernel", "local_bwd_bf16_kernel", "joint_fwd_bf16_kernel", "joint_fwd_px_kernel", "conv3x3_stream_kernel", "conv3x3_kernel", "conv3x3_pt_kernel",
            "conv3x3_wgrad_bf16_kernel", "conv3x3_wgrad_c16_kernel", "head_local_fwd_mfma_kernel", "head_local_bwd_wave_kernel",
            "head_local_bwd_fused_kernel", "bn_relu_bwd", "bn_relu_fwd")


def short(name):
    name = name.replace("void miseg::", "")
    return name.split("(")[0]


def main(src, dst):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if any(fam in k for fam in FAMILIES):
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(src, "p1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in vals:
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {"_doc": __doc__.strip(), "_command": "bash profiles/collect_pmc.sh <tag>  (python3 bench.py --steps 3 --warmup 2 under rocprofv3 --kernel-trace --pmc <group>)",
           "kernels": {}}
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mi-based-regularized-semi-supervised-segmentation_amd"))
        from miseg_amd import _cabi
        out["lib_version"] = int(_cabi.lib().miseg_version())
    except Exception as ex:  # noqa
        out["lib_version"] = None
    for k, d in sorted(vals.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"dispatches_averaged": max(len(v) for v in d.values()), "avg_us_profiled": round(sum(durs[k]) / max(1, len(durs[k])), 2)}
        e.update({c: round(v, 1) for c, v in m.items()})
        if "GRBM_GUI_ACTIVE" in m and e["avg_us_profiled"] > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / XCDS
            e["clock_ghz"] = round(cyc / e["avg_us_profiled"] / 1e3, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["issued_mfma_flop"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] * 1024.0
        for fam, fl in MFMA_FLOP:
            if fam in k and "SQ_INSTS_MFMA" in m and "local_bwd_f8" not in k:       # the f8 kernel mixes two shapes
                e["issued_mfma_flop_by_count"] = m["SQ_INSTS_MFMA"] * fl
                break
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            e["traffic_bytes_factor1"] = int((m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1000)
        out["kernels"][k] = e
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("avg_us_profiled", "mfma_busy_frac", "clock_ghz", "issued_mfma_flop", "traffic_bytes_factor1")}
                      for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
