import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
phys = BatchedPhysics(N); ids = np.arange(N)
bench.setup_envs(phys, 0, ids)
seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(10)).astype(np.float32)).to(phys.device).contiguous()
for t in range(10):
    phys.rollout(seq[t:t+1], control_steps=5)
st = phys.solver_stats().astype(np.float64) * 256
tot = st.sum(axis=1)
print("cycles per solve (last step) mean %.0f" % tot.mean(), "| shares fetch %.2f reduce %.2f compute %.2f write+barrier %.2f" % tuple(st.mean(axis=0) / tot.mean()))
print("per schedule step (assuming 100 sweeps x nsched~4.5): %.0f cycles" % (tot.mean() / 450))
