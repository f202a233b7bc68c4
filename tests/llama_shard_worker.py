"""Child process of tests/test_gpu_llama.py::test_llama_column_sharded_over_two_ranks: ONE rank of a 2-rank job whose ranks share the one GPU (gloo collectives on
CUDA tensors; RCCL refuses two ranks on one device).  Builds the same small transformers LlamaForCausalLM on every rank, shards one copy with shard_llama_layers
(BASELINE config 5's scheme: every linear column-sharded, int8-code exchange in front of o and down) and compares it with the unsharded int8 model
(swap_linears + fuse_llama_layers) on this rank.  Prints 'OK <rank> <max abs logit difference> <layers bit-identical>'."""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import transformers as tr

    import protoquant_amd as pq
    from protoquant_amd.llama import fuse_llama_layers, shard_llama_layers
    torch.manual_seed(0)
    cfg = tr.LlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
                         max_position_embeddings=256, attn_implementation="eager")
    base = tr.LlamaForCausalLM(cfg).to(torch.bfloat16).eval()          # on the CPU: the sharding copies each layer's slices to the GPU itself
    with torch.no_grad():
        for l in base.model.layers:
            l.input_layernorm.weight.copy_((1 + 0.1 * torch.randn(512)).to(torch.bfloat16)); l.post_attention_layernorm.weight.copy_((1 + 0.1 * torch.randn(512)).to(torch.bfloat16))
    ref = copy.deepcopy(base).cuda()
    pq.swap_linears(ref, fuse_gated_mlp=True)
    fuse_llama_layers(ref)
    sh = copy.deepcopy(base)
    assert shard_llama_layers(sh) == 2
    sh = sh.cuda()                                                      # embeddings, final norm, rotary tables (the sharded linears are there already)
    # the same sharding from an ALREADY-QUANTISED model (an int8 checkpoint / after swap_linears): the int8 codes and scales are sliced, nothing is re-quantised
    sh_q = copy.deepcopy(base).cuda()
    pq.swap_linears(sh_q, fuse_gated_mlp=True)
    assert shard_llama_layers(sh_q) == 2
    ids = torch.randint(0, 512, (2, 64), generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        a, b = ref(ids).logits, sh(ids).logits
        b_q = sh_q(ids).logits
    assert torch.equal(b.view(torch.int16), b_q.view(torch.int16)), "sharding a quantised model differs from sharding its float original"
    with torch.no_grad():
        # the MLP block alone, on the same replicated input: bit-identical by construction (every collective in it is exact)
        x = (torch.randn(2, 40, 512, generator=torch.Generator().manual_seed(6)) * 1.5).to(torch.bfloat16).cuda()
        hn_r, hn_s = ref.model.layers[0].post_attention_layernorm(x), sh.model.layers[0].post_attention_layernorm(x)
        m_r, m_s = ref.model.layers[0].mlp(hn_r), sh.model.layers[0].mlp(hn_s)
        # o_proj alone on a column-sharded input: this rank's heads of a replicated "attention output"
        att = (torch.randn(2, 40, 512, generator=torch.Generator().manual_seed(7))).to(torch.bfloat16).cuda()
        o_r = ref.model.layers[1].self_attn.o_proj(att)
        o_s = sh.model.layers[1].self_attn.o_proj(att[..., rank * 256:(rank + 1) * 256].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(m_r.view(torch.int16), m_s.view(torch.int16)), "sharded MLP block differs"
    assert torch.equal(o_r.view(torch.int16), o_s.view(torch.int16)), "sharded o projection differs"
    assert a.shape == b.shape == (2, 64, 512)
    diff = float((a.float() - b.float()).abs().max())
    same = bool(torch.equal(a.view(torch.int16), b.view(torch.int16)))
    # attention runs per head in stock torch ops on 4 local heads instead of 8: the same arithmetic per head; the logits agree bit for bit when the batched matmuls pick
    # the same kernels for both batch counts, and to bf16 rounding otherwise
    assert same or diff <= 0.02 * float(a.float().abs().max()), diff
    dist.barrier()
    dist.destroy_process_group()
    print(f"OK {rank} {diff:.4g} {same}", flush=True)


if __name__ == "__main__":
    main()
