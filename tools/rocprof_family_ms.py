#!/usr/bin/env python3
"""Per-family GPU time per G+D step from a `rocprofv3 --kernel-trace --stats` summary of `python bench.py ...` (the
bench line quotes it beside its own HIP-event figure -- the events of an eager re-run carry ~12 us of dispatch per launch).

    python3 tools/rocprof_family_ms.py <..._kernel_stats.csv> profiles/r06/kernel_ms_per_step.json

Families are tools/pmc_traffic.py's (= seg2eye_amd.ops.LaunchProfiler's).  Step BODIES profiled = adam_flat_kernel calls / 2 (every
timed, warm-up and per-step-timed G+D step ends with two Adam launches) + 2: the two eager G+D bodies the trainer runs before
capturing its hipGraphs launch every kernel of a step but no Adam (pix2pix_trainer._capture)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import classify, git_head, source_digest      # noqa: E402


def main():
    src, dst = sys.argv[1:3]
    rows = list(csv.DictReader(open(src)))
    adam = [int(r['Calls']) for r in rows if 'adam_flat_kernel' in r['Name']]
    steps = (max(1, adam[0] // 2) if adam else 1) + (int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    fams, other, total = defaultdict(lambda: [0, 0.0]), 0.0, 0.0
    for r in rows:
        ns = float(r['TotalDurationNs'])
        total += ns
        _, (fam, counts) = classify(r['Name'])
        if fam is None:
            other += ns
            continue
        fams[fam][1] += ns
        if counts:
            fams[fam][0] += int(r['Calls'])
    out = {'git_head': os.environ.get('S2E_GIT_HEAD') or git_head(), 'source_digest': source_digest(), 'steps_profiled': steps, 'source': os.path.basename(src),
           'kernel_ms_per_step_total': total / 1e6 / steps, 'unclassified_ms_per_step': other / 1e6 / steps,
           'families': {k: {'ms_per_step': v[1] / 1e6 / steps, 'launches_per_step': v[0] / steps} for k, v in sorted(fams.items())}}
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
