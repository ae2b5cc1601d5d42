import sys, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
free0, tot = torch.cuda.mem_get_info()
import consent_amd as ca
from consent_amd.engine import synth_host
eng = ca.Engine(ca.Params(9,4,8,2,20))
eng.run(synth_host(ca.SynthSpec.pacbio(24, 30)))
free1, _ = torch.cuda.mem_get_info()
print("device memory held by an engine after a 24-window batch: %.2f GB" % ((free0 - free1) / 1e9))
b2 = synth_host(ca.SynthSpec.pacbio(24, 30, first_window=100))
eng.run(b2); eng.run(b2)
print('third run, stage ms', {k: round(v, 2) for k, v in eng.timings().items()})
eng.close()
g.smoke()
