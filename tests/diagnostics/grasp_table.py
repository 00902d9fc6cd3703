"""Table of DESIGN.md section 8 (CPU only, oracle): cube still in the gripper after the scripted pick() with
the reference's osc.yaml gains, by solver convergence and by the recalled parameters that could weaken a grasp.
usage: python tests/diagnostics/grasp_table.py [nscenes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import grasp_study as G

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from mujoco_robot_environments_amd.model import compile as MC
    A = MC.compile_scene()
    rows = [
        ("PGS 100 sweeps (north_star)", "pgs100", {}),
        ("PGS 1000 sweeps", "pgs1000", {}),
        ("Newton (MuJoCo default)", "newton", {}),
        ("Newton, torque limits x10", "newton", {"blob": {"act_ctrlrange": np.r_[A["act_ctrlrange"][:7] * 10, A["act_ctrlrange"][7:]]}}),
        ("Newton, grip force x2", "newton", {"blob": {"grip_forcerange": A["grip_forcerange"] * 2}}),
        ("Newton, position kd 37.4 (critical damping of kp 350)", "newton", {"osc": {"kd_pos": 37.4}}),
        ("Newton, position kp/kd 525/215 (TUNED_OSC_GAINS)", "newton",
         {"osc": {"kp_pos": 525.4, "kd_pos": 215.2, "kp_ori": 766.6, "kd_ori": 158.6, "kp_null": 22.6, "kd_null": 7.2}}),
    ]
    for name, variant, ov in rows:
        print("==", name, flush=True)
        G.study(n, [variant], ov)
