"""Host-side tools that turn build artefacts / profiler output into the figures DESIGN.md quotes: their parsing is checked on small synthetic inputs (no GPU, no build)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))

ASM = """
\t.text
_Z6kernelPf:                            ; @_Z6kernelPf
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_mov_b32_e32 v1, 0
.LBB0_1:                                ; =>This Loop Header
\tglobal_load_dwordx4 v[2:5], v1, s[0:1]
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_16x16x4_f32 a[0:3], v2, v3, a[0:3]
\tv_exp_f32_e32 v6, v2
\tv_div_scale_f32 v7, vcc, v2, v3, v2
\tv_div_scale_f32 v8, vcc, v3, v3, v2
\tds_read_b128 v[8:11], v1
.LBB0_2:                                ; inner
\tv_add_f32_e32 v6, v6, v2
\ts_cbranch_scc1 .LBB0_2
\ts_add_i32 s2, s2, 1
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""


def test_isa_loops_finds_back_edges_and_counts_the_instruction_mix():
    import isa_loops as IL
    lines, loops = IL.loops_of(ASM, "_Z6kernelPf")
    assert len(loops) == 2
    outer = max(loops, key=lambda t: t[1] - t[0])
    m = IL.mix(lines[outer[0]:outer[1]])
    assert m["mfma"] == 1 and m["glob"] == 1 and m["lds"] == 1 and m["trans"] == 1 and m["div"] == 1 and m["waitcnt"] == 1
    assert m["valu"] == 4 and m["salu"] >= 2            # (the MFMA is not counted as VALU; v_exp, 2 x v_div_scale, v_add)
    inner = min(loops, key=lambda t: t[1] - t[0])
    assert IL.mix(lines[inner[0]:inner[1]])["n"] == 1


def test_pmc_traffic_applies_the_gfx950_fetch_correction(tmp_path):
    """tools/pmc_traffic.py: traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), per launch, and the per-pass sum divided by the launches of a once-per-pass kernel"""
    import csv
    import subprocess
    f, w, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "o.json"
    hdr = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
           "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]

    def rows(counter, per_kernel):
        r, i = [], 0
        for name, grid, vals in per_kernel:
            for v in vals:
                i += 1
                r.append([i, i, 0, 0, 1, 1, grid, 1, name, 256, 0, 0, 32, 0, 16, counter, v, 0, 1])
        return r
    kernels_f = [("vx_loss_finalize_k(double const*)", 64, [1.0, 1.0]), ("vx_big_k(float*)", 1024, [100.0, 100.0, 100.0, 100.0])]
    kernels_w = [("vx_loss_finalize_k(double const*)", 64, [2.0, 2.0]), ("vx_big_k(float*)", 1024, [50.0, 50.0, 50.0, 50.0])]
    for path, counter, ks in ((f, "FETCH_SIZE", kernels_f), (w, "WRITE_SIZE", kernels_w)):
        with open(path, "w", newline="") as fh:
            cw = csv.writer(fh)
            cw.writerow(hdr)
            cw.writerows(rows(counter, ks))
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "pmc_traffic.py")
    r = subprocess.run([sys.executable, tool, str(f), str(w), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out))
    big = d["kernels"]["vx_big_k"]
    assert big["launches_in_trace"] == 4
    assert big["hbm_bytes_per_launch_corrected"] == int((2 * 100.0 + 50.0) * 1024)
    assert d["kernels"]["vx_loss_finalize_k"]["hbm_bytes_per_launch_corrected"] == int((2 * 1.0 + 2.0) * 1024)
    # two passes in the trace (the once-per-pass kernel ran twice): per-pass bytes = (4 x 250 KB + 2 x 4 KB) / 2
    assert d["passes_in_trace"] == 2 and d["counter_bytes_per_pass"] == round((4 * 250 * 1024 + 2 * 4 * 1024) / 2)
