import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
print("headline", round(d["value"]), round(d["ms_per_step"], 2), d["dtype"], "parity", d["parity"]["rms_max"], "|", r["kernel"], "frac", round(r["frac"], 3), "iso", r.get("frac_isolated"), "traffic", r["traffic"])
l = d.get("fp32_mfma_kernels") or d.get("limb_kernels")
if l:
    print("other family", round(l["value"]), round(l["ms_per_step"], 2), "headline x", round(l["headline_over_this"], 3), "parity", l["parity"] and l["parity"]["rms_max"], l["parity"] and l["parity"]["ok"])
    print(json.dumps(l["roofline"])[:700])
oc = d.get("other_configs")
if oc:
    print({k: (v.get("ms_per_step") or v.get("us_per_call") or v.get("ms_per_call")) for k, v in oc.items() if isinstance(v, dict)})
    p = oc.get("dpdfnet8_48khz_hr_64_streams_1_hop", {})
    print(json.dumps(p.get("public_objects", {}))[:600]); print(json.dumps(p.get("parity_sparse")))
