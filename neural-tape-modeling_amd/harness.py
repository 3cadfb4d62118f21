"""The model-driving part of the reference's evaluation CLI (code/test-model.py:192-247, :296-418)
on in-memory segments: build the model from the weights-directory name, predict, cut INIT_LEN,
per-segment loss, mean over segments -- batched over streams and sharded over ranks instead of the
reference's BATCH_SIZE = 1 python loop (code/test-model.py:115,332)."""
import torch

from . import distributed, weights
from .model import RNN, DiffDelRNN, MRSTFTLoss, esr_dcpre_sums, esr_sums, ESR_EPS
from .utilities import nextpow2, parse_hidden_size, parse_loss, parse_model


def build_model(weight_name, max_delay_seconds=0.0, fs=44100, device="cuda", state_dict=None, warm_cache=True):
    """code/test-model.py:197-233: parse the directory name, construct, load best.pth.  The evaluation path never touches the
    parameters again, so the warm-start state is kept per parameter version (`warm_cache`, model.py's docstring)."""
    hidden_size = parse_hidden_size(weight_name)
    model_type = parse_model(weight_name)
    parse_loss(weight_name)                     # raises on a malformed name, as the reference would
    if model_type == "GRU":
        model = RNN(input_size=1, hidden_size=hidden_size, output_size=1, skip=False)
    elif model_type == "DiffDelGRU":
        max_delay_n = int(1.25 * max_delay_seconds * fs)          # code/test-model.py:223
        if max_delay_n == 0:
            max_delay_n = 2**8
        model = DiffDelRNN(input_size=1, hidden_size=hidden_size, output_size=1, skip=False,
                           max_delay=max_delay_n)
    else:
        raise SystemExit('Something is not right!')               # code/test-model.py:232
    model.load_state_dict(state_dict if state_dict is not None else weights.load_state_dict(weight_name))
    model = model.to(device).eval()
    model.warm_cache = bool(warm_cache)
    return model


def init_len(max_delay_seconds, fs=44100):
    """INIT_LEN = nextpow2(int(max_delay * fs))  (code/test-model.py:323-324)."""
    return nextpow2(int(max_delay_seconds * fs))


@torch.no_grad()
def compute_loss(model, input, target, d_traj=None, INIT_LEN=1024):
    """code/test-model.py:332-398 for the ESR entry of the loss dict.  input/target (B,1,T) on this
    rank (its shard of the segments); d_traj (B,1,T) in samples for DiffDelGRU.
    Returns the job-wide dict of distributed.reduce_loss_sums plus this rank's output tensor."""
    if isinstance(model, DiffDelRNN):
        output, _, s = model.predict_esr(input, d_traj, target, skip=INIT_LEN)   # cut first INIT_LEN samples (:367-369)
    else:
        output, s = model.predict_esr(input, target, skip=INIT_LEN)     # the ESR sums ride in the recurrent launch
    n = input.shape[-1] - INIT_LEN
    per_seg = (s[:, 0] / n) / (s[:, 1] / n + ESR_EPS)
    res = distributed.reduce_loss_sums(per_seg, s)
    res["ESR"] = res.pop("mean_segment_loss")
    # the DCPreESR entry of the loss dict (code/test-model.py:252): same aggregation on DC-blocked signals
    sd = esr_dcpre_sums(output, target, skip=INIT_LEN)
    res["DCPreESR"] = distributed.reduce_loss_sums((sd[:, 0] / n) / (sd[:, 1] / n + ESR_EPS))["mean_segment_loss"]
    # the MultiSTFT entry (code/test-model.py:253), when the segments are long enough for its largest frame
    if n > 1024:
        res["MultiSTFT"] = distributed.reduce_loss_sums(MRSTFTLoss().per_segment(output, target, skip=INIT_LEN))["mean_segment_loss"]
    return res, output


@torch.no_grad()
def apply_delay(delay, delay_trajectory, output, segment_length=None):
    """code/test-model.py:259-290 (`--ADD_DELAY` with `--MODEL GRU`), in working order: the reference
    calls `delay.init_buffer(N)` without the `max_d` argument its own class requires
    (code/model.py:326) and therefore raises TypeError; here the buffer is re-initialised to zeros with
    the delay line's current max_delay, then the trajectory is applied -- in one launch, or in the
    reference's 4096-sample chunks when `segment_length=2**12` (same numbers, state is carried)."""
    delay.init_buffer(output.shape[0], delay.max_delay)
    if segment_length is None:
        return delay(output, delay_trajectory)
    T = output.shape[-1]
    out = torch.empty(output.shape, device=output.device, dtype=torch.float32)
    for i in range(-(-T // segment_length)):
        sl = slice(i * segment_length, (i + 1) * segment_length)
        out[:, :, sl] = delay(output[:, :, sl], delay_trajectory[:, :, sl])
    return out


class BlockStreamer:
    """Block-by-block (real-time style) inference: B streams advance `block` samples per call with the GRU state kept
    on the device in preallocated buffers; one ctypes call per block on the low-latency kernel.  Measured on the
    MI355X (tools/attic/block_latency_probe.py): 20 us per 64-sample block of one stream (16 us of it kernel; the block
    lasts 1451 us at 44.1 kHz), 36 us for 16 streams x 128 samples, 132 us for 256 x 512 -- i.e. about 5 us of
    launch overhead.  Capturing the launch in a HIP graph and replaying it (`use_graph=True`, torch.cuda.CUDAGraph)
    was measured too and is SLOWER by 8 us per block: with a single kernel per block there is nothing for a graph to
    amortise, so it is off by default.  Same numbers as model.forward() on the concatenated blocks either way.

        s = BlockStreamer(model, B=16, block=128)          # RNN only; warm-start included unless warm=False
        y = s.process(x_block)                             # (B,1,block) in -> (B,1,block) view, valid until the next call
    """

    def __init__(self, model, B, block, warm=True, use_graph=False):
        from . import _lib
        from ._lib import ptr
        if not isinstance(model, RNN):
            raise TypeError("BlockStreamer drives the GRU model (RNN)")
        dev = model.GRU.weight_hh_l0.device
        self.model, self.B, self.block = model, B, block
        self.x = torch.zeros(B, 1, block, device=dev, dtype=torch.float32)
        self.y = torch.empty(B, 1, block, device=dev, dtype=torch.float32)
        if warm:
            model.initialize_hidden()
            model.warm_start()
            self.h = model.hidden.expand(1, B, model.hidden_size).contiguous().clone()
        else:
            self.h = torch.zeros(1, B, model.hidden_size, device=dev, dtype=torch.float32)
        g, o = model.GRU, model.output
        lib, variant = _lib.lib(), _lib.VARIANTS[model.kernel_variant]

        def launch():
            rc = lib.ntm_gru_forward_ex(ptr(g.weight_ih_l0), ptr(g.weight_hh_l0), ptr(g.bias_ih_l0), ptr(g.bias_hh_l0),
                                        ptr(o.weight), ptr(o.bias), model.hidden_size, ptr(self.x), ptr(self.y), B, block,
                                        block, block, ptr(self.h), variant, _lib.current_stream())
            _lib.check(rc, "ntm_gru_forward")

        self._launch = launch
        self.graph = None
        if use_graph:
            h0 = self.h.clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch()                                    # warm the code path outside the capture
            torch.cuda.current_stream().wait_stream(side)
            self.h.copy_(h0)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                launch()
            self.h.copy_(h0)                                # capture does not execute, but keep the state explicit

    @torch.no_grad()
    def process(self, x_block):
        self.x.copy_(x_block.reshape(self.B, 1, self.block), non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._launch()
        return self.y
