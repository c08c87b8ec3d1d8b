"""Turn the parity statistics a GPU test run leaves in gpurun_out/parity_stats_<pid>.json (tests/conftest.py)
into the tracked profiles/<tag>_parity_stats.json: every compared plane, plus a summary of what the
fragile-pixel / fragile-Gaussian exemptions hide.   usage: python tools/collect_parity_stats.py <tag> <file>..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, files = sys.argv[1], sys.argv[2:]
recs = []
for f in files:
    recs += json.load(open(f))
img = [r for r in recs if "frac_beyond_1e4" in r]
grd = [r for r in recs if "nonexempt_elements" in r]
summary = {
    "image_planes_compared": len(img),
    "image_values_compared": sum(r["pixels"] for r in img),
    "image_values_beyond_1e-4 (fragile pixels INCLUDED)": sum(r["beyond_1e4"] for r in img),
    "worst_plane_share_beyond_1e-4": max((r["frac_beyond_1e4"] for r in img), default=0.0),
    "worst_relative_error_any_pixel": max((r["max_rel_err"] for r in img), default=0.0),
    "fragile_pixel_share_min_max": [min((r["fragile_frac"] for r in img if r["fragile_frac"] is not None), default=None),
                                    max((r["fragile_frac"] for r in img if r["fragile_frac"] is not None), default=None)],
    "gradient_arrays_compared": len(grd),
    "gradient_rel_l2_max": max((r["rel_l2"] for r in grd), default=0.0),
    "gradient_elements_of_non_exempt_gaussians": sum(r["nonexempt_elements"] for r in grd),
    "of_which_beyond_rtol1e-3_atol1e-5": sum(r["nonexempt_beyond_strict"] for r in grd),
    "worst_array_share_beyond (non-exempt)": max((r["nonexempt_beyond_strict"] / max(r["nonexempt_elements"], 1) for r in grd), default=0.0),
    "worst_non_exempt_element_over_bar": max((r.get("nonexempt_worst_over_bar", 0.0) for r in grd), default=0.0),
    "exempt_gaussian_share_min_max": [min((r["exempt_gaussians_frac"] for r in grd), default=None),
                                      max((r["exempt_gaussians_frac"] for r in grd), default=None)],
    # the same bar with NO exemption at all (every element of every Gaussian)
    "gradient_elements_all": sum(r["pixels"] for r in grd),
    "of_which_beyond_rtol1e-3_atol1e-5 (NO exemption)": sum(r.get("all_elements_beyond_strict", 0) for r in grd),
    "worst_array_share_beyond (NO exemption)": max((r.get("all_elements_beyond_strict", 0) / max(r["pixels"], 1) for r in grd), default=0.0),
}
# per gradient array name: exempt share and misses with / without the exemption
per = {}
for r in grd:
    a = per.setdefault(r["plane"], dict(arrays=0, elements=0, beyond_no_exemption=0, nonexempt_elements=0,
                                        nonexempt_beyond=0, exempt_share_max=0.0, rel_l2_max=0.0))
    a["arrays"] += 1; a["elements"] += r["pixels"]; a["beyond_no_exemption"] += r.get("all_elements_beyond_strict", 0)
    a["nonexempt_elements"] += r["nonexempt_elements"]; a["nonexempt_beyond"] += r["nonexempt_beyond_strict"]
    a["exempt_share_max"] = max(a["exempt_share_max"], r["exempt_gaussians_frac"]); a["rel_l2_max"] = max(a["rel_l2_max"], r["rel_l2"])
summary["per_gradient_array"] = per
psnr = [r for r in recs if r.get("plane") == "psnr"]
if psnr:
    summary["psnr_fit"] = psnr[-1]
# arrays that missed the HIP-vs-oracle bars and were settled against float64 autograd (tests/test_gpu_sweep.py)
tru = [r for r in grd if "truth_rel_l2_hip" in r]
summary["gradient_arrays_settled_against_float64_autograd"] = {
    "arrays": len(tru),
    "rel_l2_vs_float64_hip_min_max": [min((r["truth_rel_l2_hip"] for r in tru), default=None),
                                      max((r["truth_rel_l2_hip"] for r in tru), default=None)],
    "rel_l2_vs_float64_oracle_min_max": [min((r["truth_rel_l2_oracle"] for r in tru), default=None),
                                         max((r["truth_rel_l2_oracle"] for r in tru), default=None)],
    "elements_beyond_bar_vs_float64_hip": sum(r["truth_miss_hip"] for r in tru),
    "elements_beyond_bar_vs_float64_oracle": sum(r["truth_miss_oracle"] for r in tru),
    "non_exempt_elements_in_those_arrays": sum(r["nonexempt_elements"] for r in tru),
}
out = os.path.join(ROOT, "profiles", "%s_parity_stats.json" % tag)
json.dump({"summary": summary, "records": recs}, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1))
