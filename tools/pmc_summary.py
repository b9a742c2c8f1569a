#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel:
   python tools/pmc_summary.py <counter_collection.csv> <COUNTER> [<COUNTER> ...] > profiles/<name>.csv
Writes kernel, launches, and per counter: total, per-launch (the counter's own unit: FETCH_SIZE / WRITE_SIZE are KB).

The kernel key is the demangled name WITHOUT its argument list: the list is cut from the right by matching the final
parenthesis, so `(anonymous namespace)::wn_block_fwd_mfma<0>(float const*, ...)` stays
`(anonymous namespace)::wn_block_fwd_mfma<0>` (round 1 split at the first "(" and collapsed every kernel of an anonymous
namespace into one row)."""
import collections
import csv
import sys


def base_name(full):
    s = full.strip()
    for suffix in (" [clone .kd]", ".kd"):
        if s.endswith(suffix):
            s = s[: -len(suffix)].rstrip()
    if s.endswith(")"):
        depth = 0
        for i in range(len(s) - 1, -1, -1):
            if s[i] == ")":
                depth += 1
            elif s[i] == "(":
                depth -= 1
                if depth == 0:
                    s = s[:i]
                    break
    if s.startswith("void "):
        s = s[5:]
    return s.strip()


def main():
    rows = csv.DictReader(open(sys.argv[1]))
    wants = sys.argv[2:]
    tot = {w: collections.defaultdict(float) for w in wants}
    cnt = {w: collections.defaultdict(int) for w in wants}
    for r in rows:
        c = r["Counter_Name"]
        if c not in tot:
            continue
        k = base_name(r["Kernel_Name"])
        tot[c][k] += float(r["Counter_Value"])
        cnt[c][k] += 1
    w = csv.writer(sys.stdout)
    head = ["kernel", "launches"]
    for c in wants:
        head += [c + "_total", c + "_per_launch"]
    w.writerow(head)
    keys = sorted(tot[wants[0]], key=lambda k: -tot[wants[0]][k])
    for k in keys:
        row = [k, cnt[wants[0]][k]]
        for c in wants:
            n = max(cnt[c][k], 1)
            row += [round(tot[c][k], 1), round(tot[c][k] / n, 1)]
        w.writerow(row)


if __name__ == "__main__":
    assert base_name("void (anonymous namespace)::wn_block_fwd_mfma<0>(float const*, float*, int) [clone .kd]") == \
        "(anonymous namespace)::wn_block_fwd_mfma<0>"
    assert base_name("void igemm::kernel<128, 128, true, 512, convop::Im2colFwd, igemm::ColPlain<4>, igemm::EpiStore>(A, B)") == \
        "igemm::kernel<128, 128, true, 512, convop::Im2colFwd, igemm::ColPlain<4>, igemm::EpiStore>"
    main()
