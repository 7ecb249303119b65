#!/usr/bin/env python3
"""One step of bench.py as a timeline, from a rocprofv3 kernel trace (`*_kernel_trace.csv`).

    python3 profiles/timeline.py profiles/r04_kernel_trace_step.csv [step]

Prints start / end / duration (us, relative to the step's first kernel) and the HSA queue of every kernel
of step `step` (default 8: a timed step, not a warm-up one) -- a step starts with k_absmax.  This is how
DESIGN.md's readings of the pyramid were made: where octave 0 ends, how long the chain of small launches
of the octaves >= 2 runs on after it, which launches share the device."""
import csv
import sys


def main():
    path = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_absmax")]
    if step + 1 >= len(starts):
        sys.exit("the trace holds %d steps" % (len(starts) - 1))
    i0, i1 = starts[step], starts[step + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i1]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3
        e = (int(r["End_Timestamp"]) - t0) / 1e3
        print("%9.1f %9.1f %8.1f  q%s %s" % (s, e, e - s, r["Queue_Id"],
                                              r["Kernel_Name"].replace("void ", "")[:60]))


if __name__ == "__main__":
    main()
