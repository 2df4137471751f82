"""Do two independent chains of dependent skinny-projection launches, captured as parallel branches of one
hipGraph, overlap each other's launch gaps?  Times one chain of 2n launches against two chains of n."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_vllm_x_amd import ops

DEV = "cuda:0"


def replay_us(g, reps=5):
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


@torch.inference_mode()
def main():
    n_layers = 28
    for m_full in (16, 30, 64, 120):
        m_half = m_full // 2
        wq = [torch.randn(4096, 1024, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(n_layers)]
        wg = [torch.randn(6144, 1024, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(n_layers)]
        xs = {m: torch.randn(m, 1024, device=DEV, dtype=torch.bfloat16) for m in (m_full, m_half)}
        outs = {}

        def chain(m, tag):
            x = xs[m]
            for l in range(n_layers):
                outs[(tag, l, 0)] = ops.linear_decode(x, wq[l], mode=0)
                outs[(tag, l, 1)] = ops.linear_decode(x, wg[l], mode=1)

        chain(m_full, "w"); chain(m_half, "w2"); torch.cuda.synchronize()
        s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream()
        g1 = torch.cuda.CUDAGraph()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0), torch.cuda.graph(g1, stream=s0):
            chain(m_full, "a")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0), torch.cuda.graph(g2, stream=s0):
            s1.wait_stream(s0)
            chain(m_half, "b")
            with torch.cuda.stream(s1):
                chain(m_half, "c")
            s0.wait_stream(s1)
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0), torch.cuda.graph(g3, stream=s0):
            chain(m_half, "d"); chain(m_half, "e")
        torch.cuda.current_stream().wait_stream(s0)
        print(f"rows {m_full}: one chain, full batch {replay_us(g1):7.1f} us | two parallel chains, half batch each {replay_us(g2):7.1f} us"
              f" | the same two chains one after the other {replay_us(g3):7.1f} us   ({2 * n_layers} launches per chain)", flush=True)


if __name__ == "__main__":
    main()
