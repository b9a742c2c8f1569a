#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the two per-kernel PMC summaries (tools/lab/pmc_step.sh):
   python tools/pmc_traffic.py profiles/r02_pmc_fetch_size_per_kernel.csv profiles/r02_pmc_write_size_per_kernel.csv > profiles/r02_pmc_traffic.json
bytes per launch = 2 x FETCH_SIZE (gfx950 tallies a 128-byte read request as 64 B: MI355X_MICROARCH, HBM/rocprofv3 section)
+ WRITE_SIZE, both reported in KB."""
import csv, json, sys


def load(path, col):
    return {r["kernel"]: float(r[col]) for r in csv.DictReader(open(path))}


fetch = load(sys.argv[1], "FETCH_SIZE_per_launch")
write = load(sys.argv[2], "WRITE_SIZE_per_launch")


def find(*needles):
    ks = [k for k in fetch if all(n in k for n in needles)]
    assert len(ks) == 1, (needles, ks)
    return ks[0]


def traffic(k):
    return int(round((2.0 * fetch[k] + write.get(k, 0.0)) * 1024))


conv = find("kernel<128, 128, true, 512", "Im2colFwdCls", "false>")         # position-class forward (stages 2-4: 8 launches / step)
conv_plain = find("kernel<128, 128, true, 512", "Im2colFwd<true>", "false>")   # dense schedule (stride-2 / 1x1 convolutions)
wn = find("wn_block_fwd_occ<0>")
print(json.dumps({
    "conv_fwd": traffic(conv), "conv_fwd_kernel": conv,
    "conv_fwd_dense_schedule": traffic(conv_plain), "conv_fwd_dense_schedule_kernel": conv_plain,
    "wn_layer": traffic(wn), "wn_layer_kernel": wn,
    "note": "bytes per launch = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, KB -> bytes, from "
            "separate rocprofv3 --pmc passes of `bench.py --no-extras` (tools/lab/pmc_step.sh -> profiles/<tag>_pmc_{fetch,write}_"
            "size_per_kernel.csv); averages over all launches of the kernel in a step (conv_fwd: the position-class forward convolutions, "
            "conv_fwd_dense_schedule: the others with Cout >= 128; wn_layer: the 20 residual-block launches, planes shrinking from 6141 to 4096 samples)"}, indent=1))
