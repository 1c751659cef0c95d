#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic_*.json from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE):
median per-launch bytes per kernel.  usage: pmc_json.py <fetch.csv> <write.csv> <out.json> [source note]"""
import csv, json, statistics as st, sys
from collections import defaultdict

STAGE = [("nl_bin_kernel", "neighbor_bin"), ("nl_fwd_kernel", "list_forward"), ("desc_rev_kernel", "descriptor_rev"),
         ("finalize_gather_kernel", "finalize"),
         ("nl_build_kernel", "neighbor_build"), ("desc_fwd_kernel", "descriptor_fwd"),
         ("gemm_nt_kernel<1", "gemm_knm"), ("gemm_nt_kernel<(GemmEpilogue)1", "gemm_knm"),
         ("gemm_nt_kernel<4", "gemm_w_covloss"), ("gemm_nt_kernel<(GemmEpilogue)4", "gemm_w_covloss"),
         ("desc_dc_kernel", "descriptor_dc"), ("desc_pair_kernel", "descriptor_pair"), ("finalize_kernel", "finalize")]


def medians(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "")
        for pat, stage in STAGE:
            if name.startswith(pat):
                acc[stage].append(float(r["Counter_Value"]) * 1024.0)  # counters are in KiB
                break
    return {k: st.median(v) for k, v in acc.items() if len(v) >= 10}


def csrc_sha():
    import glob, hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "autoforce_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "autoforce_amd", "csrc", "*.inc")) +
                    glob.glob(os.path.join(root, "autoforce_amd", "csrc", "*.h"))):
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


fetch, write = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
out = {
    "csrc_sha": csrc_sha(),  # bench.py quotes this summary only while the kernel sources are the ones it was made on
    "workload": "LiPS 4096 atoms / 512 inducing, 1 GPU",
    "source": sys.argv[4] if len(sys.argv) > 4 else "builder run",
    "unit": "bytes per launch",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh), median over the "
              "launches of the run; counters are in KiB; fetch_x2 applies the gfx950 correction of MI355X_MICROARCH.md "
              "(FETCH_SIZE reports 1/2 of a 16 B/lane coalesced stream) and is an upper bound for kernels that also "
              "issue narrower loads",
    "kernels": {k: {"fetch": fetch[k], "write": write.get(k, 0.0), "fetch_x2_plus_write": 2 * fetch[k] + write.get(k, 0.0)}
                for k in fetch},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
