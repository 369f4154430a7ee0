"""Is the encoder kernels' time cold-start (instruction cache, first touch) or steady state?  The guide alone, back to back
(pv_ivae_encode: pv_enc_l1_kernel + pv_enc_fwd_kernel), under rocprofv3 --kernel-trace --stats: compare the averages with
the same kernels' averages inside the SVI step (bench.py profile), where the decoder kernel runs in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyroved_amd as pv

torch.manual_seed(0)
m = pv.models.iVAE((28, 28), latent_dim=2, invariances=["r", "t"], device="cuda")
x = torch.rand(256, 28, 28)
for _ in range(60):
    z = m.encode(x, batch_size=256)
torch.cuda.synchronize()
print("ok", [t.shape for t in z] if isinstance(z, tuple) else z.shape)
