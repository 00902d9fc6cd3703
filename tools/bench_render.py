"""Throughput of the batched overhead camera (mre_render) at the reference's resolution.

    python tools/bench_render.py [--envs 4096] [--frames 20]
Prints one JSON line: frames/s of the whole batch, ms per batch frame (HIP events on the handle's
stream: geometry export + ray casting), and the HBM roofline of the kernel: 8 bytes are written per
pixel (f32 depth + 3 x u8 colour + u8 id), nothing else of size is read or written.
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--outputs", default="rgb,depth,seg", help="subset of rgb,depth,seg (diagnostics)")
    args = ap.parse_args()
    import torch, bench
    from mujoco_robot_environments_amd.model import compile as MC
    from mujoco_robot_environments_amd.physics import BatchedPhysics, _ptr
    from mujoco_robot_environments_amd import lib as L, rng
    N, H, W = args.envs, args.height, args.width
    phys = BatchedPhysics(N); ids = np.arange(N)
    bench.setup_envs(phys, 0, ids)
    phys.set_render_colours((rng.uniform(7, ids, [0], 12)[0].reshape(N, 4, 3) * 255).astype(np.uint8), None)
    q = np.array([0.707, 0, 0, -0.707]); Rc = MC.q2m(q / np.linalg.norm(q))
    cp = np.array([0.7, 0, 1.3], np.float32); cm = np.ascontiguousarray(Rc, np.float32).reshape(9)
    rgb = torch.empty((N, H, W, 3), dtype=torch.uint8, device=phys.device)
    depth = torch.empty((N, H, W), dtype=torch.float32, device=phys.device)
    seg = torch.empty((N, H, W), dtype=torch.uint8, device=phys.device)
    outs = set(args.outputs.split(","))
    o_rgb, o_depth, o_seg = (rgb if "rgb" in outs else None), (depth if "depth" in outs else None), (seg if "seg" in outs else None)
    def frame():
        L.check(L.lib().mre_render(phys._h, _ptr(cp), _ptr(cm), 61.0, H, W, _ptr(o_rgb), _ptr(o_depth), _ptr(o_seg), None), "mre_render")
    for _ in range(3):
        frame()
    phys.sync()
    phys.profile_enable(True)
    for _ in range(args.frames):
        frame()
    ms, n = phys.profile_read()
    per = ms / n
    bytes_written = (3.0 * ("rgb" in outs) + 4.0 * ("depth" in outs) + 1.0 * ("seg" in outs)) * N * H * W
    covered = float((seg != 255).float().mean())
    print(json.dumps({"metric": "overhead-camera frames/s (depth + rgb + segmentation)", "value": N / (per * 1e-3), "unit": "env-frames/s",
                      "envs": N, "resolution": [H, W], "ms_per_batch_frame": per,
                      "roofline": {"bound": "hbm", "achieved": bytes_written / (per * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_written / (per * 1e-3) / 8e12, "algorithmic_bytes_per_launch": bytes_written,
                                   "kernel": "mre::k_render"},
                      "pixels_hit": covered}))

if __name__ == "__main__":
    main()
