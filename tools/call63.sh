mkdir -p gpurun_out/c63
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/notch_debug.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "k_notch_scan" "$f") > $GRAFT_REPO_ROOT/gpurun_out/c63/notch_pmc_$(echo $c | tr A-Z a-z).csv; fi
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for c in ("fetch_size", "write_size"):
    rows = list(csv.DictReader(open(f"gpurun_out/c63/notch_pmc_{c}.csv")))
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    big = [v for v in vals if v > 0.5 * vals[-1]]
    print(c, len(rows), "launches; full-size:", len(big), "mean KiB", sum(big) / len(big))
PY
