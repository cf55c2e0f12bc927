#!/usr/bin/env python3
"""Kernel-level timeline of ONE synchronous host-pointer call (numpy -> numpy, 400 tiles) from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/host_call_timeline.py run
   python3 tools/host_call_timeline.py show <dir>
"""
import csv, glob, os, sys
import numpy as np


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mri_inr_amd import ModulatedSiren, synthetic as syn
    sd = syn.make_state_dict(seed=7, trained_like=True)
    t = np.random.default_rng(0).random((400, 32, 32), dtype=np.float32)
    m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    m.load_state_dict(sd); m.to("cuda")
    for _ in range(30):
        m(t)


def show(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?")), "-"))
    rows.sort()
    # the last call: from the last-but-one conv kernel of the final pair ... take the final 14 records
    idx = [i for i, r in enumerate(rows) if "encoder_conv" in r[2]]
    start = idx[-2] if len(idx) >= 2 else 0
    # back up to the copy in front of it
    while start > 0 and rows[start - 1][2].startswith("COPY") and rows[start][0] - rows[start - 1][1] < 50000:
        start -= 1
    t0 = rows[start][0]
    for s, e, n, q in rows[start:]:
        print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  q{q}  {n}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
