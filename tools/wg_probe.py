"""sliced vs atomics weight gradients on chosen shapes (diagnostics)"""
import sys, torch
sys.path.insert(0, ".")
from transformer_tts_amd import ops
torch.manual_seed(0)
for (B, t, C, N, taps, pad) in ((48, 927, 512, 80, 5, 2), (48, 927, 512, 128, 5, 2), (48, 927, 512, 80, 1, 0), (8, 927, 512, 80, 5, 2), (48, 927, 80, 512, 5, 2)):
    dy = torch.randn(B, t, N, device="cuda").bfloat16()
    x = torch.randn(B, t, C, device="cuda").bfloat16()
    res = []
    for en in (True, False, True):
        ops._WG.enabled = en
        out = torch.zeros(N, taps * C, device="cuda")
        ops.conv_wgrad(dy, x, taps, pad, out)
        torch.cuda.synchronize()
        res.append(out)
    ops._WG.enabled = True
    d = (res[0] - res[1]).abs().max().item()
    d2 = (res[0] - res[2]).abs().max().item()
    bad = ((res[0] - res[1]).abs() > 1e-2 * res[1].abs().max()).nonzero()
    print((B, t, C, N, taps), "tile", ops.lib().fs2_gemm_last_tile(), "sliced-atomics", d, "sliced-sliced", d2, "scale", res[1].abs().max().item(),
          "bad", bad.shape[0], bad[:4].tolist(), bad[-2:].tolist())
