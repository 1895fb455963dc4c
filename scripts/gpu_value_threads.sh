for t in 1 2 3 4 1 2; do
  UVC_BENCH_VALUE_THREADS=$t timeout -k 10 300 python bench.py --no-side --no-cpu-baseline --no-extras --steps 24 --warmup 4 > gpurun_out/thr_$t.json 2> gpurun_out/thr_$t.err || { tail -3 gpurun_out/thr_$t.err; exit 1; }
  python - <<PY
import json
j=json.load(open("gpurun_out/thr_$t.json"))
print("threads $t: ms/step %.3f value %.1f M  dom %s %.3f ms frac %.4f" % (j["ms_per_step"], j["value"]/1e6, j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
PY
done
