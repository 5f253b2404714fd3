R=$PWD; cd /tmp && export TMPDIR=/tmp
# the host must not run ahead of a PMC pass (every dispatch is serialised and slow): with thousands of packets queued the profiler's
# intercept queue overflowed (SIGSEGV inside a launch / "AQL packet is malformed", then a hang) — one launch at a time, and a time limit
export AMD_SERIALIZE_KERNEL=3
rm -rf $R/gpurun_out/pmc_f32
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f32 -o t -- python3 $R/tools/bench_gemm_f32.py > $R/gpurun_out/pmc_f32.log 2>&1
cd $R
python3 - <<PY
import csv, collections, re
rows = list(csv.DictReader(open("gpurun_out/pmc_f32/t_counter_collection.csv")))
agg = collections.OrderedDict()
for r in rows:
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = (re.sub(r"\(.*", "", r["Kernel_Name"])[:60], r["Grid_Size"], r.get("Workgroup_Size",""))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    if "gemm" in k[0]: print(k, n, "avg FETCH_SIZE (KB as reported) %.0f -> x2 = %.1f MB" % (v / n, 2 * v / n / 1024))
PY
