#!/bin/bash
# GPU-box helper: kernel trace of a short bench run, reduced to a per-step timeline (kernel busy time vs idle gaps).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/trace
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace/run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --tiles 4 $TRACE_ARGS > gpurun_out/trace/bench.json 2> gpurun_out/trace/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/run/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
mc = glob.glob("gpurun_out/trace/run/**/*memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy_" + r.get("Direction", "")))
ev.sort()
# last step = from the last k_prep_fast start to the end
starts = [i for i, e in enumerate(ev) if e[2] == "k_prep_fast"]
i0 = starts[-1]
# include the memsets just before
while i0 > 0 and ev[i0 - 1][2].startswith("__amd_rocclr") : i0 -= 1
seg = ev[i0:]
t0 = seg[0][0]
busy = 0; prev_end = t0; gaps = []
for s, e, n in seg:
    if s > prev_end: gaps.append((s - prev_end, n))
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
print("last step: span %.3f ms, busy %.3f ms, %d launches" % ((prev_end - t0) / 1e6, busy / 1e6, len(seg)))
for g, n in sorted(gaps, reverse=True)[:12]: print("gap %.1f us before %s" % (g / 1e3, n))
for s, e, n in seg: print("%9.1f %9.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
