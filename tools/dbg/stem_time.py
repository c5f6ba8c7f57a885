import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pips_amd import Pips, ops
m = Pips(stride=8).to("cuda:0").eval()
arena = m._packed(torch.device("cuda:0"))
rgbs = torch.randint(0, 256, (8, 3, 368, 496)).float().cuda()
for _ in range(3): ops.encoder_fwd(arena, rgbs, 8)
torch.cuda.synchronize()
