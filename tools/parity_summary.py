#!/usr/bin/env python3
"""Digest of a parity report (gpurun_out/parity_report.jsonl or profiles/parity_report_rNN.jsonl): the round-4 entries -
ensembles, IPO end-state distributions, self-envelopes, the contractive prior - one line each.
usage: python tools/parity_summary.py [report.jsonl]"""
import json
import sys


def main(path):
    for line in open(path):
        try:
            j = json.loads(line)
        except ValueError:
            continue
        t = str(j.get("test", ""))
        if t == "end_to_end_ensemble":
            m, p = j["mpjpe_mm"], j["pa_mpjpe_mm"]
            print(f"{t:28s} {j['capture']:20s} {j['math']:6s} hip {m['hip_mean']:.3f} sd {m['hip_member_sd']:.3f} (M={j['members_hip']}) "
                  f"ref {[round(v, 3) for v in m['reference']]} ref_sd {m['reference_member_sd']} half95 {m['central95_half_width']:.3f} "
                  f"mean_diff {m['mean_diff']:+.3f} resolvable {m['resolvable']:.3f} | PA max diff {p['max_diff']:.4f} sd {p['hip_member_sd']:.4f}")
        elif t == "end_to_end_ensemble_pooled":
            print(f"{t:28s} {j['math']:6s} per draw {[round(v, 3) for v in j['per_draw_mean_diff_mm']]} pooled {j['pooled_mm']:+.3f} +- {j['standard_error_mm']:.3f}")
        elif t == "ipo_end_state_distribution":
            print(f"{t:28s} {j['capture']:20s} angle {j['q_angle']['max_abs_diff']:.2e} ({j['q_angle']['max_in_member_sd']:.1f} sd) "
                  f"scale {j['q_scale']['max_abs_diff']:.2e} ({j['q_scale']['max_in_member_sd']:.1f} sd) loss {j['q_loss']['max_abs_diff']:.2e} "
                  f"({j['q_loss']['max_in_member_sd']:.1f} sd) mean loss diff {j['mean_loss_px']['diff']:+.2e} px")
        elif t.endswith("_envelope"):
            print(f"{t:28s} " + json.dumps({k: v for k, v in j.items() if k != "test" and k != "reference_runs_mm"}))
        elif t in ("configs1_ensemble", "driver_full_per_pose"):
            print(f"{t:28s} " + json.dumps({k: v for k, v in j.items() if k != "test"})[:400])
        elif t == "contractive_prior":
            print(f"{t:28s} {j['math']:6s} d_mpjpe {j['mpjpe']['d_mm']:+.5f} mm (per pose median {j['mpjpe']['per_pose_abs_median_mm']:.5f}, p99 "
                  f"{j['mpjpe']['per_pose_abs_p99_mm']:.4f}, max {j['mpjpe']['per_pose_abs_max_mm']:.3f}) d_pa {j['pa_mpjpe']['d_mm']:+.5f} mm members sd {j['hip_members_sd_mm']}")
        elif t.startswith("stage_a") or t in ("driver_cfg1", "driver_h36m_full", "driver_pw3d_full", "driver_pw3d_full_b", "driver_pw3d_full_c"):
            keys = ("d_mpjpe_mm", "d_pa_mpjpe_mm", "mpjpe_hip", "mpjpe_ref", "pa_hip", "pa_ref", "math")
            print(f"{t:28s} " + json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in j.items() if k in keys}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_report.jsonl")
