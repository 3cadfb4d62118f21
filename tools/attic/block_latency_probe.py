"""Per-block cost of real-time style inference (state on the device): plain launches vs HIP-graph replay."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
for B, block in [(1, 64), (16, 128), (64, 256), (256, 512)]:
    x = torch.rand(B, 1, block, device="cuda") - 0.5
    for use_graph in (False, True):
        s = ntm_amd.harness.BlockStreamer(m, B, block, use_graph=use_graph)
        for _ in range(20): s.process(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 500
        for _ in range(n): s.process(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B:4d} block={block:4d} graph={use_graph}: {dt*1e6:7.1f} us per block  (audio time of a block at 44.1 kHz: {block/44100*1e6:.0f} us, kernel ~{block*0.34:.0f} us)")
