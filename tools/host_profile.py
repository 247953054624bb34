"""Where the host time of a batched PointFusion step goes (cProfile over the timed loop of bench.py's workload)."""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
import bench
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 24
seqs = bench.make_sequences(list(range(B)), L, 480, 640)
frames = bench.frames_on_device(gs, seqs, torch.device("cuda"))
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = bench.run_steps(slam, gs.Pointclouds(device="cuda"), frames, None, 0, 4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pc, prev = bench.run_steps(slam, pc, frames, prev, 4, L)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(28)
