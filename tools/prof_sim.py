"""One variant of bench.py's simulator_frame leg alone, for rocprofv3: python tools/prof_sim.py separate_calls|one_call|one_call_rgb"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.simulator_leg(torch.device("cuda:0"), frames=20, warm=3, only=sys.argv[1])))
