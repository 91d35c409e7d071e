import json, sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(d["value"], d["ms_per_step"], d["step_ms_median"], d["step_ms_best"], "|", r["kernel_ms"], r["kernel_ms_median"], r["kernel_ms_min"], r.get("kernel_ms_mean_instrumented"), r["frac"], r.get("frac_of_measured_mix"), d["bit_exact"])
