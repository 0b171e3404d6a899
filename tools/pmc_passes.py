"""Generic per-kernel counter table: runs nothing itself -- summarises rocprofv3 counter passes written under one directory
(<dir>/<pass name>/**/*counter_collection.csv and *kernel_trace.csv) into one CSV row per kernel:
    python tools/pmc_passes.py <dir> <out.csv> [kernel-name substring ...]
Values are averages per dispatch; kernels are matched by substring (all kernels when none is given)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("void ", "").replace("cs::(anonymous namespace)::", "").replace("cs::", "").split("(")[0]


def main():
    prof, out, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur, ndur = collections.defaultdict(float), collections.defaultdict(int)
    counters = []
    for f in sorted(glob.glob(os.path.join(prof, "*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k, c = short(r["Kernel_Name"]), r["Counter_Name"]
            if subs and not any(s in k for s in subs):
                continue
            if c not in counters:
                counters.append(c)
            tot[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
    for f in sorted(glob.glob(os.path.join(prof, "*", "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in tot:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                ndur[k] += 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "avg_ns_under_pmc"] + counters)
        for k in sorted(tot, key=lambda k: -dur[k]):
            w.writerow([k, "%.0f" % (dur[k] / max(ndur[k], 1))] + ["%.0f" % (tot[k][c] / max(cnt[k][c], 1)) for c in counters])
    print(open(out).read())


if __name__ == "__main__":
    main()
