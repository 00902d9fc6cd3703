import sys, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.model import compile as MC
from mujoco_robot_environments_amd.tasks.rearrangement import mat2quat, home_quat
from scipy.spatial.transform import Rotation as R
from oracle import oracle as O
np.set_printoptions(precision=4, suppress=True, linewidth=200)
A = MC.compile_scene(); m = O.Model(MC.to_blob(A)); names = A['_names']['geoms']
yaw = np.deg2rad(float(sys.argv[1]) if len(sys.argv) > 1 else 30.0)
e = O.Env(m, nprops=1)
q = e.arr('qpos'); q[:7] = A['home_qpos']
q[15:22] = [0.45, 0.1, 0.4155, np.cos(yaw/2), 0, 0, np.sin(yaw/2)]
e.forward()
p = O.make_osc()
def report(tag):
    c = e.contacts()
    act = [(names[int(x[13])], names[int(x[14])], round(x[12], 5)) for x in c if x[12] < 0]
    print(tag, "cube", q[15:18], "tcp", e.arr('site_xpos')[3:6], "eef", e.arr('site_xpos')[:3], "fingers", q[7], q[11], "\n   active contacts", act[:8])
rz = abs(np.rad2deg(yaw)); rz = min([rz, rz - 90])
grasp = mat2quat(R.from_euler('xyz', [0, 180, rz], degrees=True).as_matrix())
print("grasp quat", grasp, "site quat home", mat2quat(e.arr('site_xmat')[:9].reshape(3,3)))
def run(pos, quat, grip, dur, tag):
    p.target_pos[:] = pos
    if quat is not None: p.target_quat[:] = quat
    conv = e.run_controller(p, grip, int(round(dur / 0.005)), 5)
    report(f"{tag} conv={conv}")
run([0.45, 0.1, 0.9], grasp, 0.0, 2.0, "pre-pick")
run([0.45, 0.1, 0.575], None, 0.0, 2.0, "descend")
run([0.45, 0.1, 0.575], None, 255.0, 1.0, "close")
run([0.45, 0.1, 0.9], None, 255.0, 2.0, "lift")
