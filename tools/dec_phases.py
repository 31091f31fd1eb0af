import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv=[sys.argv[0]]
import bench
from tools.gemm_sweep import time_graph
c=bench.CFG
model=bench.build_model(c, torch.device("cuda")); eng=model.engine()
feats,depth,lab,dur,tgt=bench.make_inputs(c, torch.device("cuda"), 1)
eng.forward(feats,depth,lab,"train",True); torch.cuda.synchronize()
w=eng.last["w"]
def run(): eng._decoder_fused(w, lab, True, 1/0.9)
prev=0
for stop in list(range(1,11))+[0]:
    os.environ["R3D_DEC_STOP"]=str(stop)
    t=time_graph(run, reps=10, replays=5)
    print(f"stop={stop:2d} cumulative {t:7.2f} us  (+{t-prev:6.2f})", flush=True); prev=t
