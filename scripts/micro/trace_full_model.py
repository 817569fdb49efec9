import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GSN_CHAIN_TRACE"] = "1"
import torch, bench
dev = torch.device("cuda", 0)
step, G = bench.full_model_closure(dev, 16384)
step(); torch.cuda.synchronize()
