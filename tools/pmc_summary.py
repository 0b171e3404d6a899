"""Per-kernel summaries of the rocprofv3 counter passes of tools/profile.sh:  python tools/pmc_summary.py <prof dir> <out prefix>"""
import collections
import csv
import glob
import sys

prof, out = sys.argv[1], sys.argv[2]
PROOFS = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # proofs inside each profiled bench.py run (warmup + steps)


def short(name):
    return name.replace("void ", "").replace("cs::(anonymous namespace)::", "").replace("cs::", "").split("(")[0]


def collect(sub, counters):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(float)
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (prof, sub)):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c not in counters:
                continue
            k = short(r["Kernel_Name"])
            tot[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
    for f in glob.glob("%s/%s/*/*kernel_trace.csv" % (prof, sub)):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, cnt, dur


ft, fc, _ = collect("pmc_FETCH_SIZE", {"FETCH_SIZE"})
wt, wc, _ = collect("pmc_WRITE_SIZE", {"WRITE_SIZE"})
with open(out + "_hbm_traffic_pmc.csv", "w", newline="") as f:
    wcsv = csv.writer(f)
    wcsv.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_per_dispatch", "WRITE_SIZE_KB_per_dispatch", "fetch_x2_plus_write_GB_per_dispatch",
                   "fetch_x2_plus_write_GB_per_proof"])
    for k in sorted(ft, key=lambda k: -ft[k]["FETCH_SIZE"]):
        n = max(fc[k]["FETCH_SIZE"], 1)
        fe = ft[k]["FETCH_SIZE"] / n
        wr = wt[k]["WRITE_SIZE"] / max(wc[k]["WRITE_SIZE"], 1)
        wcsv.writerow([k, n, "%.0f" % fe, "%.0f" % wr, "%.4f" % ((2 * fe + wr) * 1024 / 1e9), "%.4f" % ((2 * fe + wr) * 1024 / 1e9 * n / PROOFS)])

names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
vt, vc, vd = collect("pmc_valu", set(names))
with open(out + "_valu_pmc.csv", "w", newline="") as f:
    wcsv = csv.writer(f)
    wcsv.writerow(["kernel", "dispatches", "avg_ns_under_pmc"] + [n + "_per_dispatch" for n in names])
    for k in sorted(vt, key=lambda k: -vd[k]):
        n = max(vc[k][names[0]], 1)
        wcsv.writerow([k, n, "%.0f" % (vd[k] / n)] + ["%.0f" % (vt[k][c] / max(vc[k][c], 1)) for c in names])
print("wrote", out + "_hbm_traffic_pmc.csv", out + "_valu_pmc.csv")
