"""Summarise tools/r6_c3_state.sh: per PMC pass the mean counter values of the main kernel's full-size dispatches."""
import collections, csv, glob, sys
base = sys.argv[1]
vals = collections.defaultdict(list)
for f in sorted(glob.glob(base + "/pmc*/*/*_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "kolb_pool" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        m = max(v)
        big = [x for x in v if x > 0.5 * m] if m > 0 else v
        vals[k] = sum(big) / len(big)
g = lambda k: vals.get(k, float("nan"))
for k in sorted(vals):
    print("  %-36s %.4g" % (k, vals[k]))
print("  mean TCP->TCC read latency  %.0f cycles (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
print("  mean TCC->EA read latency   %.0f cycles (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ)" % (g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum")))
print("  UTCL1 translation miss rate %.4f of %.3g requests" % (g("TCP_UTCL1_TRANSLATION_MISS_sum") / g("TCP_UTCL1_REQUEST_sum"), g("TCP_UTCL1_REQUEST_sum")))
print("  L2 hit rate                 %.4f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
