import sys, os, torch
sys.path.insert(0, os.getcwd())
from wekws_amd.frontend import Fbank
from wekws_amd.utils import synth
pcm = torch.from_numpy(synth.synth_pcm(1024, 16000, seed=3)).cuda()
fb = Fbank(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
for _ in range(6): f = fb(pcm)
torch.cuda.synchronize()
print(tuple(f.shape))
