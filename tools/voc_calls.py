"""Aggregate a rocprofv3 --kernel-trace CSV of tools/vocoder_bench.py by (kernel, grid): usage voc_calls.py <kernel_trace.csv> <out.txt>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not any(s in n for s in ("conv1d", "respair", "act_split", "tconv", "dwconv", "mean3", "stft", "mel", "layernorm", "fsq")):
        continue
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void fs::", "").replace("fs::", "")
    key = (short, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as out:
    tot = sum(sum(v) for v in agg.values())
    out.write("total %.1f us\n" % (tot / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.write("%-44s grid %-9s y %-3s n=%-3d total=%9.1f us avg=%8.1f us\n" % (k[0], k[1], k[2], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3))
