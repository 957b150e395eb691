"""The spurious validation words of DESIGN section 6 ("open issue"): which transport of the four flag words shows them?

usage: QAGNN_PREP_OVERLAP=0 python scripts/r5_race_variants.py <variant>     (variant: base | ring | clone | sync)
  base   the shipped transport: a fresh pinned buffer per call, copy_(non_blocking) from the flag words, an event
  ring   pinned buffers from a ring allocated once (no pinned allocation / free on the hot path)
  clone  device clone of the flag words first, then the copy from the clone
  sync   a device synchronisation in front of every replay's copy (timing control: does the fault need the overlap?)
Instead of raising, a bad word is reported together with the DEVICE words read after a full synchronisation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qagnn_amd._lib as L

variant = sys.argv[1] if len(sys.argv) > 1 else 'base'
W = L.ERR_WATCH
bad = []
ring, ring_i = None, 0


def host_buf():
    global ring, ring_i
    if variant != 'ring':
        return torch.empty(4, dtype=torch.int32, pin_memory=True)
    if ring is None:
        ring = torch.zeros(256, 4, dtype=torch.int32).pin_memory()
    ring_i = (ring_i + 1) % 256
    return ring[ring_i]


def push(self, flags, what, reset, tag):
    src = flags.clone() if variant == 'clone' else flags
    if variant == 'sync' and tag == 'replay':
        torch.cuda.synchronize()
    host = host_buf()
    host.copy_(src, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    self.pending.append((ev, host, what, reset, flags, src, tag))


def watch(self, flags, what, reset=None):
    if torch.cuda.is_current_stream_capturing():
        if self.sink is not None:
            self.sink.append((flags, what, reset))
        return
    push(self, flags, what, reset, 'eager')


def after_replay(self, watched):
    for flags, what, reset in watched:
        push(self, flags, what, reset, 'replay')


def poll(self, block=False):
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return
    keep = []
    for item in self.pending:
        ev, host, what, reset, flags, src, tag = item
        if block:
            ev.synchronize()
        if not ev.query():
            keep.append(item)
            continue
        w = host.tolist()
        if any(w):
            torch.cuda.synchronize()
            again = host.tolist()
            bad.append(tag)
            whole = torch.empty(0, dtype=torch.int32, device=flags.device).set_(flags.untyped_storage())
            o = flags.storage_offset()
            if len(bad) <= 2:
                print(f'    around the flag words: n_chunks {whole[o - 4:o].tolist()} err[0:16] {whole[o:o + 16].tolist()} the 8 words behind (cnt_s) '
                      f'{whole[o + 16:o + 24].tolist()} nonzero words in the 64 KB in front {int((whole[o - 16384:o - 4] != 0).sum())} of 16380', flush=True)
            print(f'BAD[{tag}] host words {[hex(x & 0xffffffff) for x in w]} (re-read after sync {[hex(x & 0xffffffff) for x in again]}) '
                  f'device words now {flags.tolist()} host buffer {hex(host.data_ptr())} flags at {hex(flags.data_ptr())} :: {what[:60]}', flush=True)
    self.pending = keep


L._ErrWatch.watch, L._ErrWatch.after_replay, L._ErrWatch.poll = watch, after_replay, poll
import bench
sys.argv = ['bench.py', '--steps', '5', '--warmup', '2', '--repeats', '1', '--no-cpu-baseline', '--no-pmc', '--no-configs']
try:
    bench.main()
except Exception as e:
    print('bench raised', type(e).__name__, str(e)[:200])
W.poll(block=True)
print(f'== variant {variant} env PREP_OVERLAP={os.environ.get("QAGNN_PREP_OVERLAP")} DEV_KERNARG={os.environ.get("HIP_FORCE_DEV_KERNARG")} '
      f'SDMA={os.environ.get("HSA_ENABLE_SDMA")} LIB={os.path.basename(os.environ.get("QAGNN_LIB", "default"))} ' + ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('DEBUG_')) + f': {len(bad)} bad reads {bad}', flush=True)
