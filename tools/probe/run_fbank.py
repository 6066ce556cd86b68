import sys, os, torch
sys.path.insert(0, os.getcwd())
from wekws_amd.frontend import Fbank
from wekws_amd.utils import synth
pcm = torch.from_numpy(synth.synth_pcm(1024, 16000, seed=3)).cuda()
fb = Fbank(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
# (1,200 calls: the trace's average is a steady-state one -- the review of round 5 found six calls, still inside the clock ramp)
for _ in range(int(os.environ.get('FBANK_CALLS', '1200'))): f = fb(pcm)
torch.cuda.synchronize()
print(tuple(f.shape))
