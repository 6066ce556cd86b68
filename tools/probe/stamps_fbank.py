import sys, os, torch
sys.path.insert(0, os.getcwd())
from wekws_amd.frontend import Fbank
from wekws_amd.utils import synth
pcm = torch.from_numpy(synth.synth_pcm(1024, 16000, seed=3)).cuda()
fb = Fbank(40)
for _ in range(3): f = fb(pcm)
torch.cuda.synchronize()
d = f.flatten()[:8].cpu().numpy()
names = ["dc", "preemph+win", "fft", "power", "log+store", "mel", "-", "loop-top"]
print(" ".join(f"{n}={int(v)}" for n, v in zip(names, d)), "total", int(d.sum()))
