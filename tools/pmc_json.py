#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic_*.json from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE):
median per-launch bytes per kernel.  usage: pmc_json.py <fetch.csv> <write.csv> <out.json> [source note] [sq.csv kernel_trace.csv]
With the SQ pass (SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU, ... of tools/collect_profiles.sh) and its kernel
trace: per-kernel issue statistics, and `desc_valu_frac` = the share of the SIMDs' cycles in which the two descriptor
kernels (list + forward, reverse) had a vector instruction executing."""
import csv, json, os, statistics as st, sys
from collections import defaultdict

STAGE = [("nl_bin_kernel", "neighbor_bin"), ("nl_fwd_kernel", "list_forward"), ("desc_rev_kernel", "descriptor_rev"),
         ("finalize_gather_kernel", "finalize"), ("finalize_next_kernel", "finalize_bin_next"),
         ("gemm_nt_kernel8<1", "gemm_knm"), ("gemm_nt_kernel8<(GemmEpilogue)1", "gemm_knm"),
         ("gemm_nt_kernel8r64<1", "gemm_knm"), ("gemm_nt_kernel8r64<(GemmEpilogue)1", "gemm_knm"),
         ("nl_build_kernel", "neighbor_build"), ("desc_fwd_kernel", "descriptor_fwd"),
         ("gemm_nt_kernel<1", "gemm_knm"), ("gemm_nt_kernel<(GemmEpilogue)1", "gemm_knm"),
         ("gemm_nt_kernel<4", "gemm_w_covloss"), ("gemm_nt_kernel<(GemmEpilogue)4", "gemm_w_covloss"),
         ("gemm_nt_kernel8r64<4", "gemm_w_covloss"), ("gemm_nt_kernel8r64<(GemmEpilogue)4", "gemm_w_covloss"),
         ("gemm_nt_kernel8<4", "gemm_w_covloss"), ("gemm_nt_kernel8<(GemmEpilogue)4", "gemm_w_covloss"),
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
    "workload": os.environ.get("PMC_WORKLOAD", "LiPS 4096 atoms / 512 inducing, 1 GPU"),
    "source": sys.argv[4] if len(sys.argv) > 4 else "builder run",
    "unit": "bytes per launch",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh), median over the "
              "launches of the run; counters are in KiB; fetch_x2 applies the gfx950 correction of MI355X_MICROARCH.md "
              "(FETCH_SIZE reports 1/2 of a 16 B/lane coalesced stream) and is an upper bound for kernels that also "
              "issue narrower loads",
    "kernels": {k: {"fetch": fetch[k], "write": write.get(k, 0.0), "fetch_x2_plus_write": 2 * fetch[k] + write.get(k, 0.0)}
                for k in fetch},
}
if len(sys.argv) > 6:
    # SQ_ACTIVE_INST_* count quad-cycles summed over all waves (MI355X_MICROARCH.md): x4 = SIMD cycles with such an
    # instruction executing; the chip has 256 CUs x 4 SIMDs; kernel cycles = duration x the clock the pass held
    # (GRBM_GUI_ACTIVE is not in this pass: 2.4 GHz nominal is used and stated)
    sq = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sys.argv[5])):
        name = r["Kernel_Name"].replace("void ", "")
        for pat, stage in STAGE:
            if name.startswith(pat):
                sq[stage][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    dur = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[6])):
        name = r["Kernel_Name"].replace("void ", "")
        for pat, stage in STAGE:
            if name.startswith(pat):
                dur[stage].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                break
    CLK, SIMDS = 2.4e9, 1024
    issue = {}
    for stage, c in sq.items():
        if stage not in dur or len(dur[stage]) < 10:
            continue
        med = {k: st.median(v) for k, v in c.items()}
        cyc = st.median(dur[stage]) * 1e-9 * CLK * SIMDS
        waves = None
        issue[stage] = {
            "duration_us_in_this_pass": round(st.median(dur[stage]) * 1e-3, 2),
            "valu_busy_frac": round(4 * med.get("SQ_ACTIVE_INST_VALU", 0.0) / cyc, 3),
            "issue_busy_frac": round(4 * med.get("SQ_ACTIVE_INST_ANY", 0.0) / cyc, 3) if "SQ_ACTIVE_INST_ANY" in med else None,
            "wave_wait_mem_frac": round(med.get("SQ_WAIT_ANY", 0.0) / max(med.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
            "wave_wait_issue_frac": round(med.get("SQ_WAIT_INST_ANY", 0.0) / max(med.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
            "insts_valu": med.get("SQ_INSTS_VALU"), "insts_salu": med.get("SQ_INSTS_SALU"), "insts_lds": med.get("SQ_INSTS_LDS"),
            "lds_bank_conflict_frac": round(med["SQ_LDS_BANK_CONFLICT"] / max(med.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 3)
            if "SQ_LDS_BANK_CONFLICT" in med else None,
        }
    out["issue"] = issue
    dk = [k for k in ("list_forward", "descriptor_rev") if k in issue]
    if dk:
        tot = sum(issue[k]["duration_us_in_this_pass"] for k in dk)
        out["desc_valu_frac"] = round(sum(issue[k]["valu_busy_frac"] * issue[k]["duration_us_in_this_pass"] for k in dk) / tot, 3)
        out["desc_issue_frac"] = round(sum((issue[k]["issue_busy_frac"] or 0.0) * issue[k]["duration_us_in_this_pass"] for k in dk) / tot, 3)
        out["desc_valu_note"] = ("SQ pass of tools/collect_profiles.sh: 4 x SQ_ACTIVE_INST_VALU / (kernel duration x 2.4 GHz x 1024 SIMDs), "
                                 "time-weighted over nl_fwd and desc_rev; desc_issue_frac the same with SQ_ACTIVE_INST_ANY: these kernels "
                                 "are bound by instruction issue and dependent latencies, not by HBM bytes")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
