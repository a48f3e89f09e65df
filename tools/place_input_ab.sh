cd "$GRAFT_REPO_ROOT"
for v in 1 0 1 0; do
  LSDR_BENCH_PLACE_INPUT=$v LSDR_BENCH_MORE_ONLY=c3,anf1,c2_offset LSDR_BENCH_FULL=gpurun_out/ab_in_$v.json timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/ab_in.err
  python - <<PY
import json
j=json.load(open("gpurun_out/ab_in_$v.json"))
m=j["more"]
print("place_input=$v", {k:(m[k].get("value"), m[k].get("pass"), m[k].get("error","")[:200]) for k in m})
PY
done
