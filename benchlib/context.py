"""benchlib.context — stock torch-ROCm legs on the same GPU (context beside the headline line, not the contract's baseline)."""
import torch

from .common import graph_of


def gpu_context(x, wq, ws, y_ref, med_of):
    """Context, not the contract's baseline: the SAME qlinear written with stock torch-ROCm ops around `torch._int_mm` on THIS GPU — what
    the reference's Python would execute on an MI355X (hipBLASLt int8 GEMM with int32 output, eager elementwise kernels around it) —
    plus the int8 GEMM alone and the un-quantised bf16 linear.  hipGraph replays (no host gaps), HIP events, median.  The eager pipeline's
    result is compared with the library's y (QSPEC written in torch ops; see the note on torch-ROCm's division below)."""
    wt = wq.t()

    def pipeline():
        xf = x.float()
        amax = xf.abs().amax(dim=1, keepdim=True)
        s = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
        xq = torch.round(xf / s).clamp_(-127, 127).to(torch.int8)
        acc = torch._int_mm(xq, wt)
        return ((acc.float() * s) * ws).to(torch.bfloat16)
    out = {}
    try:
        yp = pipeline()
        # torch-ROCm's elementwise float division on the GPU is not the correctly rounded one (a few per cent of the row scales differ in
        # the last bit from the CPU's and from this library's, which matches the CPU bit for bit), so a small share of outputs differs
        out["outputs_differing_from_library"] = int((yp.view(torch.int16) != y_ref.view(torch.int16)).sum().item())
        out["outputs"] = int(yp.numel())
        out["max_abs_diff"] = float((yp.float() - y_ref.float()).abs().max().item())
        w_bf16 = (wq.float() * ws[:, None]).to(torch.bfloat16)
        xq0 = torch.round(x.float()).clamp_(-127, 127).to(torch.int8)
        legs = {"torch_rocm_int8_pipeline_us": pipeline, "torch_int_mm_alone_us": lambda: torch._int_mm(xq0, wt),
                "torch_bf16_linear_us": lambda: torch.nn.functional.linear(x, w_bf16)}
        n = 5
        graphs = {k: graph_of(f, n) for k, f in legs.items()}
        for g in graphs.values():
            g.replay()
        torch.cuda.synchronize()
        ts = {k: [] for k in legs}
        for _ in range(9):
            for k, g in graphs.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); b.synchronize()
                ts[k].append(a.elapsed_time(b) * 1e3 / n)
        for k, v in ts.items():
            out[k] = round(med_of(v), 2)
        out["what"] = ("stock torch-ROCm ops on this GPU, hipGraph replays: the QSPEC pipeline around torch._int_mm (hipBLASLt int8, int32 out) = what the "
                       "reference's Python would run here; that GEMM alone; the bf16 linear the int8 path replaces")
    except Exception as e:      # context must never lose the main line
        out["error"] = str(e)[:300]
    return out

