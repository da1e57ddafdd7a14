"""Debug/bench aid: the prefill flash-attention kernel at the 7B shape (S = 2048, 32 heads, head_dim 128)."""
import sys, time, torch
sys.path.insert(0, ".")
from exllama_amd import cuda_ext as ce
ext = ce.exllama_ext
dev = "cuda:0"
S, H, D = 2048, 32, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, S, H * D, device=dev, generator=g).half()
kc = torch.randn(1, H, S, D, device=dev, generator=g).half()
vc = torch.randn(1, H, S, D, device=dev, generator=g).half()
out = torch.empty_like(q)
lib = ext._lib
def run():
    ce.check(lib.exl_attention(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), None, 1, S, H, H, D, S, 0, None,
                               torch.cuda.current_stream().cuda_stream), "attention")
run(); torch.cuda.synchronize()
qq = q.view(1, S, H, D).transpose(1, 2).float()
ref = torch.nn.functional.scaled_dot_product_attention(qq, kc.float(), vc.float(), is_causal=True).transpose(1, 2).reshape(1, S, H * D)
print("max err", (out.float() - ref).abs().max().item(), "scale", ref.abs().max().item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"flash prefill S={S}: {us:.1f} us  {4 * S * S * H * D / 2 / us / 1e6:.1f} TFLOP/s (causal flops)")
