"""CPU oracle: torque saturation and Cartesian tracking during the scripted descent."""
import sys, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.model import compile as MC
from mujoco_robot_environments_amd.tasks.rearrangement import mat2quat
from scipy.spatial.transform import Rotation as R
from oracle import oracle as O
np.set_printoptions(precision=3, suppress=True, linewidth=220)
A = MC.compile_scene(); m = O.Model(MC.to_blob(A))
x, y, rz = 0.4888, 0.1797, 44.7
e = O.Env(m, nprops=1)
q = e.arr('qpos'); q[:7] = A['home_qpos']
q[15:22] = [x, y, 0.4155, np.cos(np.deg2rad(rz)/2), 0, 0, np.sin(np.deg2rad(rz)/2)]
e.forward()
p = O.make_osc()
grasp = mat2quat(R.from_euler('xyz', [0, 180, rz], degrees=True).as_matrix())
lim = np.array([87, 87, 87, 87, 12, 12, 12.0])
def run(pos, quat, dur, tag, every=5):
    p.target_pos[:] = pos
    if quat is not None: p.target_quat[:] = quat
    print(tag)
    for k in range(int(round(dur / 0.005))):
        tau = e.osc(p)
        if k % every == 0:
            eef = e.arr('site_xpos')[:3]
            print("  t %.3f eef %s tau %s sat %s" % (k * 0.005, eef, tau, (np.abs(tau) > lim).astype(int)))
        e.run_controller(p, 0.0, 1, 5)
run([x, y, 0.9], grasp, 2.0, "pre-pick", every=40)
run([x, y, 0.575], None, 0.6, "descend", every=4)
print("cube", q[15:18])
