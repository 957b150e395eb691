"""Autograd operators of the hot path, each a thin differentiable wrapper over C-ABI kernels.

The kernel provider is `qagnn_amd._lib.HipKernels` (libqagnn_hip.so on the current HIP stream).  There is no
CPU implementation in this package: `kernels()` raises if the library or the GPU is missing.  Tests may install a
torch emulation of the same kernel interface with `set_kernels()` to exercise the HOST logic (packing, autograd
wiring, the hand-derived backward formulas) on a machine without a GPU; that emulation lives under tests/.

Internal layout ("head-padded"): a feature row of d = H*dh floats is stored as H groups of HP = roundup4(dh)
floats, DP = H*HP per row, pads are zero.  All matrices handed to the kernels are contiguous 2-D fp32.
"""
import math

import torch
from torch.amp import custom_bwd, custom_fwd

# every operator computes in fp32: inside torch.autocast (the reference's --fp16 mode, qagnn.py:254-257) inputs are cast back
_fwd = custom_fwd(device_type='cuda', cast_inputs=torch.float32)
_bwd = custom_bwd(device_type='cuda')

_K = None


def kernels():
    global _K
    if _K is None:
        from ._lib import HipKernels
        _K = HipKernels()
    return _K


def set_kernels(k):
    """Install a kernel provider (tests only); returns the previous one."""
    global _K
    old, _K = _K, k
    return old


H_HEADS = 4  # GATConvE hard-codes head_count=4 (reference modeling_qagnn.py:387)

# bias / type-table gradients as rows of weight-gradient products that are computed anyway (type indicators in S's padding, a column
# of ones in relu(bn(h1))) wherever the layout has a padding column for them; the stand-alone column reductions remain for layouts
# without padding (dim_per_head a multiple of 4).  (A/B of round 3: profiles/r3_run9_byproduct_grads_ab.txt)
BYPRODUCT_GRADS = True


def roundup(x, m):
    return (x + m - 1) // m * m


class HeadLayout:
    """Index maps between the reference's dense feature axis (d) and the head-padded axis (DP)."""

    def __init__(self, d, device):
        assert d % H_HEADS == 0, 'emb_dim must be divisible by the 4 attention heads'
        self.d, self.dh = d, d // H_HEADS
        self.HP = roundup(self.dh, 4)
        if self.HP > 64:
            raise NotImplementedError(f'dim_per_head={self.dh} > 64 is outside what the edge kernels support')
        self.DP = H_HEADS * self.HP
        j = torch.arange(self.DP)
        h, i = j // self.HP, j % self.HP
        self.dense_pos = torch.nonzero(i < self.dh).flatten().to(device)  # [d] padded position of dense index k (ascending)
        # a padding column (exactly 0 in every activation, zero row / column in every packed weight), or -1 when dh needs no padding:
        # where relu(bn(h1)) is made to carry a column of ones (see GatMlpFn)
        self.ones_col = self.dh if (self.HP > self.dh and BYPRODUCT_GRADS) else -1

    def pad(self, x):
        """[*, d] -> [*, DP] (zeros in the pads): one constant-pad of the [*, H, dh] view (its backward is a slice, no scatter)."""
        lead = x.shape[:-1]
        return torch.nn.functional.pad(x.reshape(*lead, H_HEADS, self.dh), (0, self.HP - self.dh)).reshape(*lead, self.DP).contiguous()

    def unpad(self, xp):
        lead = xp.shape[:-1]
        return xp.reshape(*lead, H_HEADS, self.HP)[..., :self.dh].reshape(*lead, self.d)


class _PlanGatherFn(torch.autograd.Function):
    """outputs = split(cat(sources, 0)[idx]);  backward = K inverse gathers (no atomics, fixed order)."""

    @staticmethod
    @_fwd
    def forward(ctx, plan, *sources):
        K = plan.kernels_for(sources)
        if K is not None:  # one launch over the tensors where they lie (qagnn_gather_multi_f32)
            packed = K.gather_multi([t.contiguous() for t in sources], plan.src_tid, plan.src_off)
        else:
            flat = torch.cat([t.reshape(-1) for t in sources] + [plan.zeros[:1]])  # (the appended zero element: the plan's cached zeros, no fill)
            packed = flat.index_select(0, plan.idx)
        ctx.plan = plan
        ctx.src_shapes = [t.shape for t in sources]
        ctx.set_materialize_grads(False)  # operands without a gradient arrive as None, not as ~50 freshly zero-filled tensors
        return tuple(packed[a:a + n].view(shape) for a, n, shape in plan.slices)

    @staticmethod
    @_bwd
    def backward(ctx, *grads):
        plan = ctx.plan
        join_wgrads(next((g for g in grads if g is not None), None))  # the packed operands' gradients may still be in flight
        K = plan.kernels_for(grads)
        if K is not None and any(g is not None for g in grads):  # one launch: every source element sums its packed copies' gradients
            gsrc = K.gather_multi_sum([None if g is None else g.contiguous() for g in grads], plan.inv_tid, plan.inv_off)
            out, off = [], 0
            for shape in ctx.src_shapes:
                n = 1
                for v in shape:
                    n *= v
                out.append(gsrc[off:off + n].view(shape))
                off += n
            return (None,) + tuple(out)
        parts = []
        z = plan.zeros  # cached zeros: operands that received no gradient (the non-transposed weight copies) cost no fill kernel
        for g, (a, n, shape), n4 in zip(grads, plan.slices, plan.padded):
            parts.append(g.reshape(-1) if g is not None else z[:n])
            if n4 != n:
                parts.append(z[:n4 - n])
        parts.append(z[:1])  # the slot missing multiplicities point to
        gflat = torch.cat(parts)
        gsrc = gflat.index_select(0, plan.inv[0])
        for k in range(1, len(plan.inv)):
            gsrc = gsrc + gflat.index_select(0, plan.inv[k])
        out, off = [], 0
        for shape in ctx.src_shapes:
            n = 1
            for v in shape:
                n *= v
            out.append(gsrc[off:off + n].view(shape))
            off += n
        return (None,) + tuple(out)


class GatherPlan:
    """Pack many parameters into the kernels' operand layouts with ONE gather (and a few inverse gathers in backward).

    The packing of a weight (slice, transpose, head-pad, concatenate) only moves elements.  The readable torch
    packing code is therefore run ONCE on tensors holding element ids; the ids that come out are the gather index.
    Afterwards a forward costs `torch.cat(sources)` + `index_select` instead of ~10 tiny kernels per weight, and the
    backward gathers, for every source element, the gradients of the (at most K) packed positions it was copied to --
    no atomics, fixed summation order.
    """

    def __init__(self):
        self.sig = None

    def _build(self, sources, build):
        dev = sources[0].device
        ids, off = [], 1
        for t in sources:
            ids.append((torch.arange(t.numel(), dtype=torch.float64, device=dev) + off).view(t.shape))
            off += t.numel()
        outs = build(ids)
        total = off - 1  # index of the appended zero element
        idx, self.slices, self.padded, pos = [], [], [], 0
        for o in outs:
            i = o.reshape(-1).round().long() - 1
            i = torch.where(i < 0, torch.full_like(i, total), i)
            n = i.numel()
            n4 = roundup(n, 4)  # keep every packed operand 16-byte aligned
            if n4 != n:
                i = torch.cat([i, torch.full((n4 - n,), total, dtype=torch.long, device=dev)])
            idx.append(i)
            self.slices.append((pos, n, tuple(o.shape)))
            self.padded.append(n4)
            pos += n4
        self.idx = torch.cat(idx)
        self.dtype = sources[0].dtype
        self.zeros = torch.zeros(max(self.padded) + 1, dtype=self.dtype, device=dev)
        # inverse map: for source element s, the packed positions holding a copy of it (missing -> position `pos`)
        order = torch.sort(self.idx, stable=True).indices
        counts = torch.bincount(self.idx, minlength=total + 1)[:total]
        starts = torch.cumsum(counts, 0) - counts
        K = int(counts.max().item()) if total > 0 else 1
        self.inv = []
        for k in range(max(K, 1)):
            take = order[torch.clamp(starts + k, max=order.numel() - 1)]
            self.inv.append(torch.where(counts > k, take, torch.full_like(take, pos)))
        # the same maps for the one-launch kernels (qagnn_gather_multi{,_sum}_f32): (tensor, element) per position, -1 = zero
        def locate(flat_idx, starts, lens):
            """flat position in a virtual concatenation with slice k at [starts[k], starts[k + 1]) and lens[k] real elements"""
            st = torch.tensor(starts, dtype=torch.long, device=dev)
            k = torch.bucketize(flat_idx, st[1:], right=True).clamp_(max=len(lens) - 1)
            o = flat_idx - st[k]
            ok = (flat_idx >= 0) & (flat_idx < starts[-1]) & (o < torch.tensor(lens, dtype=torch.long, device=dev)[k])
            return torch.where(ok, k, torch.full_like(k, -1)).to(torch.int32).contiguous(), torch.where(ok, o, torch.zeros_like(o)).to(torch.int32).contiguous()
        src_starts, a = [0], 0
        for t in sources:
            a += t.numel()
            src_starts.append(a)
        self.src_tid, self.src_off = locate(self.idx, src_starts, [t.numel() for t in sources])
        inv = torch.stack(self.inv)
        self.inv_tid, self.inv_off = locate(inv, [a for a, _, _ in self.slices] + [pos], [n for _, n, _ in self.slices])
        self.n_sources, self.fits = len(sources), total < 2 ** 31 - 1 and pos < 2 ** 31 - 1

    def kernels_for(self, tensors):
        """the provider's one-launch gather, where it has one and the tables fit its kernel arguments"""
        if not GATHER_FUSED or not self.fits:
            return None
        K = kernels()
        lim = getattr(K, 'GATHER_MAX', 0)
        if self.n_sources > lim or len(self.slices) > lim:
            return None
        ts = [t for t in tensors if t is not None]
        if not ts or any(t.dtype != torch.float32 or t.device != ts[0].device for t in ts):
            return None  # mixed dtypes / devices (autocast, a half-precision parameter): the cat + index_select path promotes, the kernel cannot
        return K if (ts[0].is_cuda or K.name != 'hip') else None

    def __call__(self, sources, build):
        sig = tuple((tuple(t.shape), str(t.device), t.dtype) for t in sources)
        if sig != self.sig:
            self._build(sources, build)
            self.sig = sig
        return list(_PlanGatherFn.apply(self, *sources))


# ------------------------------------------------------------------------------------------------------------------
import os as _os

_SIDE_STREAMS = {}
# (Bias gradients as a by-product of the weight-gradient GEMM's k-loop -- the colsum_groups form of qagnn_gemm_tn_colsum_f32 -- measured
# slower than the separate column sums, profiles/r1_run21_fused_colsum_ab.txt, and is no longer wired into the operators.)


# ------------------------------------------------------------------------------------------------------------------
# Weight-gradient GEMMs under the edge backward (QAGNN_WGRAD_OVERLAP).
#
# The edge backward kernels are bound by the row-gather rate of the texture path (time = gathers x ~33 us at E' = 460 800,
# profiles/r1_run56_*): they leave the matrix cores idle.  The weight-gradient GEMMs are MFMA-bound, independent of the
# data-gradient chain, and per layer they take about as long as the edge backward (~430 us vs ~417 us).  So inside the
# stack they are not launched where autograd reaches them: they are QUEUED, and the whole queue is issued on a second HIP
# stream right before the next edge backward starts on the main stream.  Unlike a fork/join inside one operator (run 19: a GEMM
# could only ever overlap another GEMM, measured -4.4 %, removed), the join is deferred to the one consumer of these
# gradients, GatherPlan's backward (plus an end-of-backward engine callback as a safety net).  Rules that make this safe:
#   * only operators created inside `wgrad_scope()` defer, and the stack guarantees that every weight operand there comes
#     straight out of GatherPlan, so nothing on the main stream reads a deferred gradient before the join;
#   * a gradient consumed inside the graph is either computed on the main stream (the grouped column reduction over dC when
#     `BYPRODUCT_GRADS` is off) or -- the default -- handed out as a VIEW of a deferred weight-gradient product (the node-type-table
#     gradient = rows [tab_col, tab_col + T) of S^T dC, LinearNNFn.tabcol) whose ONE consumer, SplitColsFn.backward, joins the side
#     stream before it reads; hop() refuses the combination "table gradient as a by-product" + "tables computed inside the hop"
#     (torch.addmm would read the view without a join);
#   * outputs are allocated on the main stream up front; every input of a queued launch is kept alive until the join, so
#     the caching allocator cannot hand its memory to a main-stream kernel while the side stream still reads it.
WGRAD_OVERLAP = _os.environ.get('QAGNN_WGRAD_OVERLAP', '1') == '1'
_DEFER = [False]


class wgrad_scope:
    """Operators built inside this scope may defer their weight-gradient launches (see above)."""

    def __enter__(self):
        self.prev = _DEFER[0]
        _DEFER[0] = WGRAD_OVERLAP
        if not self.prev:
            reset_wgrad_queue()  # a backward that raised after defer_wgrads() must not leak its jobs into this pass
        return self

    def __exit__(self, *exc):
        _DEFER[0] = self.prev
        return False


# QAGNN_WGRAD_POISON=1 (tests): deferred outputs start as NaN, so a reader that runs before the queued launch shows up
WGRAD_POISON = _os.environ.get('QAGNN_WGRAD_POISON', '0') == '1'


def _wg_empty(ref, shape):
    t = ref.new_empty(shape)
    return t.fill_(float('nan')) if WGRAD_POISON else t


class _WgradQueue:
    n_deferred = 0      # launches queued so far (tests)
    pending = []        # closures that launch weight-gradient kernels into pre-allocated outputs
    keep = []           # tensors those launches read: alive until the join
    unjoined = None     # (main stream, side stream) with queued work that the main stream has not waited for
    callback_queued = False


def _end_of_backward():
    _WgradQueue.callback_queued = False
    join_wgrads()


def reset_wgrad_queue():
    """Drop whatever an aborted backward left behind (called when a new forward opens a wgrad_scope).

    If a backward raises after defer_wgrads(), the engine's final callback does not run: `callback_queued` would stay True (so
    the end-of-backward join is never re-armed) and the stale closures would be issued -- into freed outputs -- by the next
    backward.  Work already issued on the side stream is still joined, so its buffers cannot be recycled under it."""
    q = _WgradQueue
    q.pending = []
    q.callback_queued = False
    if q.unjoined is not None:
        q.unjoined[0].wait_stream(q.unjoined[1])
        q.unjoined = None
    q.keep.clear()


def defer_wgrads(jobs, keep):
    """Queue weight-gradient launches (called from inside a backward)."""
    q = _WgradQueue
    q.n_deferred += len(jobs)
    q.pending.extend(jobs)
    q.keep.extend(t for t in keep if t is not None)
    if not q.callback_queued:  # whatever happens, the queue is issued and joined before backward() returns
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        q.callback_queued = True


def flush_wgrads(ref=None):
    """Issue everything queued on the side stream, forked from the current (main) stream; returns at once."""
    q = _WgradQueue
    if not q.pending:
        return
    jobs, q.pending = q.pending, []
    ref = ref if ref is not None else (q.keep[0] if q.keep else None)
    if ref is not None and not ref.is_cuda:  # host-logic tests (torch emulation of the kernels): no streams, same order
        for job in jobs:
            job()
        return
    main = torch.cuda.current_stream(ref.device if ref is not None else None)  # the stream of the device holding the operands
    if q.unjoined is not None and q.unjoined[0] != main:
        q.unjoined[0].wait_stream(q.unjoined[1])
    key = main.device_index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=main.device)
    side = _SIDE_STREAMS[key]
    side.wait_stream(main)  # the queued launches read tensors the main stream produced
    with torch.cuda.stream(side):
        for job in jobs:
            job()
    q.unjoined = (main, side)


def join_wgrads(ref=None):
    """Issue what is still queued, then make the main stream wait for the side stream; releases the kept inputs."""
    q = _WgradQueue
    flush_wgrads(ref)
    if q.unjoined is not None:
        q.unjoined[0].wait_stream(q.unjoined[1])
        q.unjoined = None
    q.keep.clear()


# ------------------------------------------------------------------------------------------------------------------
# Graph preparation under the input GEMM (QAGNN_PREP_OVERLAP).  The graph orderings depend on the batch's integer inputs
# only; their ~13 launches are short, latency-bound integer kernels (0.28 ms per batch) while the stage in front of the stack
# is one MFMA-bound gather-GEMM (entity table x cpt_transform, ~0.28 ms).  QAGNN.forward therefore issues the preparation on a
# second stream first and joins it right before the stack.  No autograd node runs on that stream; the graph storage is
# allocated there and only read elsewhere after the join (and freed after the step, i.e. before the next fork).
PREP_OVERLAP = _os.environ.get('QAGNN_PREP_OVERLAP', '1') == '1'
_PREP_STREAMS = {}


def build_graph(adj, node_type, n_etype, n_ntype, block_n):
    """The batch's graph orderings from whatever the caller handed over as `adj`: a data_utils.PackedGraphBatch (load-time
    blobs, one device buffer) or the reference's (edge_index [2, E], edge_type [E]) int64 pair."""
    K = kernels()
    from .data_utils import PackedGraphBatch
    packed = adj if isinstance(adj, PackedGraphBatch) else (adj[0] if isinstance(adj[0], PackedGraphBatch) else None)
    if packed is not None:
        assert packed.n == block_n and packed.n_etype == n_etype and packed.n_ntype == n_ntype, 'blob store built for another model shape'
        return K.graph_from_blobs(packed, node_type)
    return K.graph_prep(adj[0], adj[1], node_type, n_etype, n_ntype, block_n=block_n)


def graph_prep_async(adj, node_type, n_etype, n_ntype, block_n):
    """-> (graph, join): the preparation is enqueued on the side stream; call join() on the consumer stream before using it."""
    if not (PREP_OVERLAP and node_type.is_cuda):
        return build_graph(adj, node_type, n_etype, n_ntype, block_n), (lambda: None)
    dev = node_type.device
    main = torch.cuda.current_stream(dev)
    key = main.device_index
    if key not in _PREP_STREAMS:
        _PREP_STREAMS[key] = torch.cuda.Stream(device=dev)
    side = _PREP_STREAMS[key]
    side.wait_stream(main)  # the inputs (and last step's readers of the recycled storage) are ordered before the fork
    with torch.cuda.stream(side):
        graph = build_graph(adj, node_type, n_etype, n_ntype, block_n)
    return graph, (lambda: torch.cuda.current_stream(dev).wait_stream(side))


# ------------------------------------------------------------------------------------------------------------------
# B operands of the large NN products, packed once per forward.  The bf16-split NN kernel wants its weight operand split into three
# bf16 images in LDS order (csrc/gemm_nn2.hip); done per product that is one extra launch in front of each of the ~39 large NN
# products of a step (0.18 ms of kernel time and as many host calls at 320 subgraphs).  The stack therefore packs ALL of them with one
# launch right behind the operand-packing gather and registers them with the library, which recognises a registered operand by its
# pointers.  `owner` (a module) holds the packed buffer and the weights until its next forward replaces them: a registered pointer
# always names live, unchanged memory.
# (module attributes, not environment switches: the A/B runs that set them are recorded in profiles/r4_run28_round4_switches_ab.txt;
# the tests still flip GATHER_FUSED to compare the one-launch gather with the cat + index_select form it replaces)
GATHER_FUSED = True     # GatherPlan through qagnn_gather_multi{,_sum}_f32
HEAD_FUSED = True       # the head behind the pooling as two kernels each way
PREPACK_MIN_ROWS = 8192  # = the library's threshold for packed B images (csrc/gemm_nn2.hip: nn2_packed_ok)


_PREPACK_STATE = None  # weakref.WeakKeyDictionary: owner module -> [registry tag, what must stay alive until the next forward]
_prepack_tags = [0]


def prepack_weights(owner, pairs, rows):
    """pairs = [(B1n, B2n or None), ...] in their [No, K] layouts, as the products of this forward (and its backward) will pass them.

    The per-owner state (registry tag, packed buffer, pinned weights) lives OUTSIDE the module, keyed weakly by the owner object: a
    copy.deepcopy / pickle of the model carries none of it, a copy registers under a tag of its own on its first forward, and the tag
    is cleared when its owner is collected (the tag is a process-wide counter -- id() values are recycled, a counter is not)."""
    global _PREPACK_STATE
    K = kernels()
    if not (rows >= PREPACK_MIN_ROWS and pairs and pairs[0][0].is_cuda and hasattr(K, 'prepack')):
        return
    import weakref
    if _PREPACK_STATE is None:
        _PREPACK_STATE = weakref.WeakKeyDictionary()
    st = _PREPACK_STATE.get(owner)
    if st is None:
        _prepack_tags[0] += 1
        st = _PREPACK_STATE[owner] = [(1 << 62) | _prepack_tags[0], None]
        weakref.finalize(owner, K.prepack_clear, st[0])
    st[1] = K.prepack(pairs, st[0])  # (replaces, and thereby releases, what the previous forward registered)


class GradAcc:
    """Running gradient of ONE tensor that several operators of the stack read (the score embedding S: every hop; the stack
    input: hop 0 and the output GEMM).  Autograd would give each reader its own [N, .] gradient and add them with elementwise
    kernels; here the readers' data-gradient GEMMs accumulate into one buffer in place (the GEMM epilogue's accumulate flag)
    and only the reader whose backward runs LAST -- the one that was created first, the stack is a chain -- returns the total.
    One instance per forward pass."""
    __slots__ = ('buf',)

    def __init__(self):
        self.buf = None


def _acc_grad(acc, last, compute):
    """compute(out, accumulate) -> gradient tensor (written into `out` when given).  Returns what the operator hands to autograd."""
    if acc is None:
        return compute(None, False)
    if acc.buf is None:
        acc.buf = compute(None, False)
    else:
        compute(acc.buf, True)
    if not last:
        return None
    total, acc.buf = acc.buf, None  # handed over: a second backward through the same graph starts a fresh total
    return total


class LinearNNFn(torch.autograd.Function):
    """C = [A1|A2] @ [B1t;B2t] + bias + rowtab[rowidx]   (B*t are [K, No] = W^T; B* are the same weights as [No, K]).
    acc = (acc1, last1, acc2, last2) or None: GradAcc of A1 / A2 and whether this operator returns their totals."""

    @staticmethod
    @_fwd
    def forward(ctx, A1, B1t, B1, A2, B2t, B2, bias, rowtab, rowidx, acc, tabcol=-1):
        K = kernels()
        # operand maxima left by the producers of A1 / A2 (amax_note): the product then runs in the three-MFMA form, and so do its
        # data-gradient and weight-gradient products when the incoming gradient carries its maximum too
        am1, am2 = (amax_lookup(A1), amax_lookup(A2)) if _wants_amax(K, A1) else (None, None)
        if am1 is None or (A2 is not None and am2 is None):
            am1 = am2 = None
        kw = dict(a_amax1=am1, a_amax2=am2) if am1 is not None else {}
        C = K.gemm_nn(A1, B1t, A2, B2t, bias=bias, rowtab=rowtab, rowidx=rowidx, B1n=B1, B2n=B2, **kw)
        ctx.am = (am1, am2)
        ctx.save_for_backward(A1, B1, A2, B2, rowidx, B1t, B2t)
        ctx.has = (bias is not None, rowtab is not None, rowtab.size(0) if rowtab is not None else 0)
        # tabcol >= 0: columns [tabcol, tabcol + G) of A2 hold the indicators of rowidx (see type_indicators): the row-table gradient
        # is then rows [tabcol, tabcol + G) of A2^T dC, a by-product of the weight-gradient GEMM
        ctx.tabcol = tabcol if (A2 is not None and rowtab is not None) else -1
        ctx.defer = _DEFER[0]
        ctx.acc = acc if acc is not None else (None, False, None, False)
        return C

    @staticmethod
    @_bwd
    def backward(ctx, dC):
        K = kernels()
        A1, B1, A2, B2, rowidx, B1t, B2t = ctx.saved_tensors
        amc = amax_lookup(dC) if _wants_amax(K, dC) else None  # (max |dC| from dC's producer, e.g. GeluDropoutFn.backward)
        if not dC.is_contiguous():
            dC, amc = dC.contiguous(), None
        need = ctx.needs_input_grad
        has_bias, has_tab, G = ctx.has
        dbias = drowtab = None
        want_tab, want_bias = has_tab and need[7], has_bias and need[6]
        am1, am2 = getattr(ctx, 'am', (None, None))
        nn_kw = dict(a_amax1=amc) if amc is not None else {}
        tn_h2 = amc is not None and am1 is not None and (A2 is None or am2 is not None)
        if ctx.defer:  # weight gradients queued for the next edge backward (see defer_wgrads)
            jobs, dB1t, dB2t = [], None, None
            if need[1] and A2 is not None and need[4]:
                # both weight gradients share dC: ONE split-K launch and one chunk sum into one [K1 + K2, No] buffer (qagnn_gemm_tn2_f32)
                joint = _wg_empty(dC, (A1.size(1) + A2.size(1), dC.size(1)))
                dB1t, dB2t = joint[:A1.size(1)], joint[A1.size(1):]
                jobs.append((lambda: K.gemm_tn_h2(A1, dC, am1, amc, A2=A2, amax_a2=am2, out=joint)) if tn_h2 else (lambda: K.gemm_tn2(A1, A2, dC, out=joint)))
            else:
                if need[1]:
                    dB1t = _wg_empty(dC, (A1.size(1), dC.size(1)))
                    jobs.append(lambda: K.gemm_tn(A1, dC, out=dB1t))
                if A2 is not None and need[4]:
                    dB2t = _wg_empty(dC, (A2.size(1), dC.size(1)))
                    jobs.append(lambda: K.gemm_tn(A2, dC, out=dB2t))
            if want_tab and ctx.tabcol >= 0 and dB2t is not None and not want_bias:
                drowtab = dB2t[ctx.tabcol:ctx.tabcol + G]  # (deferred with dB2t: its consumer, SplitColsFn.backward, joins the side stream)
            elif want_tab:  # consumed inside the graph (table GEMM backward): stays on the main stream
                drowtab = K.colsum(dC, rowidx, G)
                if want_bias:
                    dbias = drowtab.sum(0)
            elif want_bias:
                cs = _wg_empty(dC, (1, dC.size(1)))
                jobs.append(lambda: K.colsum(dC, out=cs))
                dbias = cs[0]
            defer_wgrads(jobs, (A1, A2, dC))
            acc1, last1, acc2, last2 = ctx.acc
            dA1 = dA2 = None
            if need[0]:
                if acc1 is None and A2 is None and dC.size(0) <= 2048 and dC.size(1) >= 512 and dC.size(0) % 4 == 0:
                    dA1 = K.gemm_tn(dC.t().contiguous(), B1)  # few rows, long reduction: see below
                else:
                    dA1 = _acc_grad(acc1, last1, lambda out, accu: K.gemm_nn(dC, B1, out=out, accumulate=accu, B1n=B1t, **nn_kw))
            if A2 is not None and need[3]:
                dA2 = _acc_grad(acc2, last2, lambda out, accu: K.gemm_nn(dC, B2, out=out, accumulate=accu, B1n=B2t, **nn_kw))
            return dA1, dB1t, None, dA2, dB2t, None, dbias, drowtab, None, None, None
        cs = None
        joint = None
        if need[1] and A2 is not None and need[4]:  # (see the deferred path)
            joint = K.gemm_tn_h2(A1, dC, am1, amc, A2=A2, amax_a2=am2) if tn_h2 else K.gemm_tn2(A1, A2, dC)
        dB1t = joint[:A1.size(1)] if joint is not None else (K.gemm_tn(A1, dC) if need[1] else None)
        tab_from_wgrad = want_tab and not want_bias and ctx.tabcol >= 0 and A2 is not None and need[4]
        if (want_tab or want_bias) and not tab_from_wgrad:
            cs = K.colsum(dC, rowidx if want_tab else None, G if want_tab else 1)
        if joint is not None:
            dB2t = joint[A1.size(1):]
        else:
            dB2t = K.gemm_tn(A2, dC) if (A2 is not None and need[4]) else None
        if want_tab and ctx.tabcol >= 0 and dB2t is not None and cs is None:
            cs = dB2t[ctx.tabcol:ctx.tabcol + G]
        if want_tab:
            drowtab = cs
            if want_bias:
                dbias = cs.sum(0)
        elif want_bias:
            dbias = cs[0]
        acc1, last1, acc2, last2 = ctx.acc
        if need[0] and acc1 is None and A2 is None and dC.size(0) <= 2048 and dC.size(1) >= 512 and dC.size(0) % 4 == 0:
            # few rows, long reduction (the class tables: 612 x 2080): an NN launch would be 5 blocks walking 130 k-tiles one
            # after the other; the split-K weight-gradient kernel computes the same product as (dC^T)^T B1 in parallel chunks
            dA1 = K.gemm_tn(dC.t().contiguous(), B1)
        else:
            dA1 = _acc_grad(acc1, last1, lambda out, accu: K.gemm_nn(dC, B1, out=out, accumulate=accu, B1n=B1t, **nn_kw)) if need[0] else None
        dA2 = None
        if A2 is not None and need[3]:
            dA2 = _acc_grad(acc2, last2, lambda out, accu: K.gemm_nn(dC, B2, out=out, accumulate=accu, B1n=B2t, **nn_kw))
        return dA1, dB1t, None, dA2, dB2t, None, dbias, drowtab, None, None, None


def linear_nn(A1, B1t, B1, A2=None, B2t=None, B2=None, bias=None, rowtab=None, rowidx=None, acc=None, tabcol=-1):
    return LinearNNFn.apply(A1, B1t, B1, A2, B2t, B2, bias, rowtab, rowidx, acc, tabcol)


def type_indicators(S, ntype, col0, T):
    """S [N, SP] with exactly-zero padding columns from col0 on -> a copy whose columns [col0, col0 + T) hold the one-hot of ntype.
    The matching rows of every weight that multiplies S are zero padding, so no product changes; but rows [col0, col0 + T) of
    S^T dC -- computed anyway, as part of the weight gradient -- are then the per-type column sums of dC, i.e. the gradient of the
    node-type table TT that the projection adds by node type.  That retires a grouped column reduction over dKMQ [N, 3 DP] per layer
    (41 us + a 19-27 us final stage at 64 000 rows).  The gradient that flows back into S is zeroed at the indicator positions by
    scatter's own backward."""
    assert col0 + T <= S.size(1)
    word = amax_lookup(S)
    # the kernels clamp a node type outside [0, T) to T - 1 and report it through ERR_WATCH: the indicator must follow the same rule,
    # or an invalid id would land in a later padding column (or past SP) and dTT would silently lose that row's contribution
    out = S.scatter(1, (ntype.clamp(0, T - 1).view(-1, 1) + col0), 1.0)
    if word is not None:  # max |.| of the copy = max(max |S|, 1.0): bit patterns of non-negative floats order like integers
        amax_note(out, torch.clamp(word, min=0x3F800000))
    return out


class SplitColsFn(torch.autograd.Function):
    """[R, k*W] -> k contiguous [R, W] blocks (one copy); backward = one stack + one copy instead of k slice-backward
    (zeros + copy) pairs.  Used for the per-layer tables that the stack computes for all k layers with one GEMM."""

    @staticmethod
    def forward(ctx, X, k):
        R, W = X.size(0), X.size(1) // k
        ctx.shape = (R, k, W)
        return tuple(X.view(R, k, W).transpose(0, 1).contiguous().unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        R, k, W = ctx.shape
        ref = next(g for g in grads if g is not None)
        join_wgrads(ref)  # the type-table gradients are rows of deferred weight-gradient products (LinearNNFn.tabcol)
        gs = [g if g is not None else torch.zeros_like(ref) for g in grads]
        return torch.stack(gs, 1).reshape(R, k * W), None


def split_cols(X, k):
    return SplitColsFn.apply(X, k)


# ------------------------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------------------------
# Operand maxima for the three-MFMA GEMM form (csrc/gemm_nn2.hip), where a tensor is produced by one operator and read by another: the
# producer leaves max |.| in a 4-word int32 tensor and NOTES it against the tensor object; the consumer (the natively sequenced stack)
# looks it up and hands the word to the library (qagnn_hop_args.x_amax / s_amax), which then skips its own reduction pass.  An entry is
# valid for exactly the tensor object it was noted for, at the version it had (an in-place op invalidates it); anything else -- no entry,
# a dead tensor, a view, a modified tensor -- is a miss, and a miss only costs the reduction pass.
_AMAX_NOTES = {}


def amax_note(t, word):
    import weakref
    key = id(t)
    _AMAX_NOTES[key] = (weakref.ref(t, lambda _r, k=key: _AMAX_NOTES.pop(k, None)), t._version, word)
    return t


def amax_lookup(t):
    e = _AMAX_NOTES.get(id(t)) if t is not None else None
    if e is None or e[0]() is not t or t._version != e[1]:
        return None
    return e[2]


def _wants_amax(K, X):
    """the provider runs the three-MFMA form and X is large enough for it (the library's own threshold: csrc/hop.hip, hop_h2)"""
    return getattr(K, 'gemm_split', 1) >= 2 and getattr(K, 'name', '') == 'hip' and X.dim() == 2 and X.size(0) >= 8192


class GeluDropoutFn(torch.autograd.Function):
    """Y = dropout(gelu_tanh(X), p)   (utils/layers.py:10-14 + F.dropout); the keep mask is regenerated in backward."""

    @staticmethod
    @_fwd
    def forward(ctx, X, p, seed):
        ctx.save_for_backward(X)
        ctx.p, ctx.seed = p, seed
        K = kernels()
        if _wants_amax(K, X):
            Y, word = K.gelu_dropout_fwd(X, p, seed, amax=True)
            return amax_note(Y, word)
        return K.gelu_dropout_fwd(X, p, seed)

    @staticmethod
    @_bwd
    def backward(ctx, dY):
        (X,) = ctx.saved_tensors
        K = kernels()
        if _wants_amax(K, X):  # (max |dX| rides on the pass: the Linear in front of this GELU runs its backward products in the three-MFMA form)
            dX, word = K.gelu_dropout_bwd(X, dY.contiguous(), ctx.p, ctx.seed, amax=True)
            return amax_note(dX, word), None, None
        return K.gelu_dropout_bwd(X, dY.contiguous(), ctx.p, ctx.seed), None, None


_seed_counter = [0]


_rank_salt = [None]


def _rank():
    """Rank of this process in the default process group (0 outside torch.distributed): ranks that were seeded identically
    must still draw different dropout masks for their different shards."""
    if _rank_salt[0] is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            _rank_salt[0] = dist.get_rank()
        else:
            return int(_os.environ.get('RANK', '0'))  # not cached: the group may be initialised later
    return _rank_salt[0]


def manual_seed(seed):
    """Seed everything a run's dropout masks depend on: torch's generators, this module's per-process call counter and -- on every
    visible GPU -- the library's device-side seed EPOCH (the word a replayed hipGraph advances, qagnn_seed_epoch_advance; it is
    device state that torch.manual_seed does not know about, and it is not part of a checkpoint: a resumed run that wants the masks
    of the original run calls manual_seed again and replays from there)."""
    torch.manual_seed(seed)
    _seed_counter[0] = 0
    if torch.cuda.is_available() and _K is not None and getattr(_K, 'name', '') == 'hip':
        for d in range(torch.cuda.device_count()):
            with torch.cuda.device(d):
                _K.seed_epoch_set(0)


def next_seed():
    """Per-call dropout seed from torch's CPU seed (so torch.manual_seed controls it), a per-process call counter and the rank."""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _seed_counter[0] * 0x85EBCA77 + _rank() * 0xC2B2AE3D27D4EB4F) % (2 ** 63)


def gelu_dropout(X, p, training):
    p = float(p) if training else 0.0
    return GeluDropoutFn.apply(X, p, next_seed() if p > 0 else 0)


# ------------------------------------------------------------------------------------------------------------------
class SinBasisFn(torch.autograd.Function):
    """B[r, j] = sin(js[j] * score[r]), j < J, zero-padded to `ldo` columns (reference modeling_qagnn.py:69-71).  The reference
    differentiates through it (node scores are data there, but a caller may learn them): dscore[r] = sum_j dB[r, j] js[j] cos(.)."""

    @staticmethod
    @_fwd
    def forward(ctx, score, js, ldo):
        ctx.save_for_backward(score, js)
        return kernels().sin_basis(score, js, ldo)

    @staticmethod
    @_bwd
    def backward(ctx, dB):
        score, js = ctx.saved_tensors
        arg = score.reshape(-1, 1) * js.reshape(1, -1)
        return (dB[:, :js.numel()] * torch.cos(arg) * js.reshape(1, -1)).sum(1).view_as(score), None, None


def sin_basis(score, js, ldo):
    if score.requires_grad:
        return SinBasisFn.apply(score, js, ldo)
    return kernels().sin_basis(score, js, ldo)  # the usual case (scores are inputs): no autograd node


# ------------------------------------------------------------------------------------------------------------------
class EdgeAttnFn(torch.autograd.Function):
    """aggr = edge attention + aggregation (see qagnn_edge_attn_fwd_f32); also returns the un-scaled attention a."""

    @staticmethod
    @_fwd
    def forward(ctx, KMQ, EkEm, graph, HP, qscale):
        aggr, a, alpha = kernels().edge_attn_fwd(graph, KMQ, EkEm, HP, qscale)
        ctx.save_for_backward(KMQ, EkEm, a, alpha)
        ctx.graph, ctx.HP, ctx.qscale = graph, HP, qscale
        ctx.mark_non_differentiable(a)
        ctx.set_materialize_grads(False)  # `a` gets no gradient: without this autograd zero-fills an [E', 4] tensor per layer and step
        return aggr, a

    @staticmethod
    @_bwd
    def backward(ctx, G, _da):
        if G is None:
            return None, None, None, None, None
        KMQ, EkEm, a, alpha = ctx.saved_tensors
        flush_wgrads(G)  # queued weight-gradient GEMMs go out on the side stream now: they run under the gather-bound kernels
        dKMQ, dEkEm = kernels().edge_attn_bwd(ctx.graph, KMQ, EkEm, ctx.HP, ctx.qscale, a, alpha, G.contiguous())
        return dKMQ, dEkEm, None, None, None


def edge_attention(KMQ, EkEm, graph, HP, qscale):
    return EdgeAttnFn.apply(KMQ, EkEm, graph, HP, qscale)


# ------------------------------------------------------------------------------------------------------------------
class GatMlpFn(torch.autograd.Function):
    """GATConvE.mlp + activation of one hop, fused:  X' = dropout(gelu(Lin2(relu(BN(Lin1(aggr))))))

    reference: modeling_qagnn.py:443 (mlp, def :408), :48-49 (GELU, dropout).  BatchNorm1d runs over ALL N rows
    (PAD rows included); in training mode it uses biased batch statistics and updates the running buffers with the
    unbiased variance (momentum 0.1), exactly like torch.nn.BatchNorm1d.  BN + ReLU are folded into the operand load
    of the second GEMM (and of its weight-gradient GEMM), so relu(bn(h1)) is never materialised.
    """

    @staticmethod
    @_fwd
    def forward(ctx, aggr, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean, run_var, training, eps, p, seed, apply_act, running,
                row_weight, ones_col=-1):
        K = kernels()
        R = aggr.size(0)
        if training and row_weight is None and _colstats_ok(K, R, aggr.size(1), W1t.size(1)):
            # batch statistics as a by-product of the first Linear's GEMM epilogue (per-tile partials) + ONE launch that combines
            # them and does the BatchNorm bookkeeping, instead of two more passes over h1 and five launches
            h1, part = K.gemm_nn(aggr, W1t, bias=b1, B1n=W1, colstats=True)
            mean, var, invstd, scale, shift = K.bn_stats_finalize(part, R, gamma, beta, eps, running, ones_col).unbind(0)
        else:
            h1 = K.gemm_nn(aggr, W1t, bias=b1, B1n=W1)
            if training:
                sc = 1.0 / R if row_weight is None else 1.0  # row_weight [R] (sums to 1): weighted statistics
                mean = K.colsum(h1, scale=sc, roww=row_weight)[0]
                var = K.colvar_sum(h1, mean, scale=sc, roww=row_weight)  # biased
            else:
                mean, var = run_mean, run_var
            # invstd / scale / shift and (train mode) the module's running-statistics update: one launch
            invstd, scale, shift = K.bn_finalize(mean, var, gamma, beta, eps, running, ones_col)
        out = K.gemm_nn(h1, W2t, bias=b2, a_scale=scale, a_shift=shift, B1n=W2)
        y = K.gelu_dropout_fwd(out, p, seed) if apply_act else out
        ctx.ones_col = ones_col
        ctx.save_for_backward(aggr, h1, out, mean, invstd, scale, shift, W1, W2, gamma, row_weight, W1t, W2t)
        ctx.cfg = (training, p, seed, R, apply_act)
        ctx.defer = _DEFER[0]
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)  # (no zero-filled stand-ins for the gradients of mean / var)
        return y, mean, var

    @staticmethod
    @_bwd
    def backward(ctx, dy, _dm, _dv):
        if dy is None:
            return (None,) * 19
        K = kernels()
        aggr, h1, out, mean, invstd, scale, shift, W1, W2, gamma, row_weight, W1t, W2t = ctx.saved_tensors
        training, p, seed, R, apply_act = ctx.cfg
        dout = K.gelu_dropout_bwd(out, dy.contiguous(), p, seed) if apply_act else dy.contiguous()
        oc = ctx.ones_col  # >= 0: relu(bn(h1)) carries a column of ones there, so row oc of dW2t = relu(bn(h1))^T dout IS colsum(dout)
        db2 = K.colsum(dout)[0] if oc < 0 else None
        if ctx.defer:  # weight gradients of both Linears queued for the next edge backward (see defer_wgrads)
            Cc = dout.size(1)
            dW2t = _wg_empty(dout, (h1.size(1), Cc))
            if oc >= 0:
                db2 = dW2t[oc]  # deferred with dW2t; its only reader is GatherPlan's backward, which joins the side stream
            dr = K.gemm_nn(dout, W2, B1n=W2t)
            red = K.bn_bwd_reduce(dr, h1, mean, invstd, scale, shift)
            dh1, db1 = K.bn_relu_bwd_colsum(dr, h1, mean, invstd, scale, shift, gamma, red, 1.0 / R if training else 0.0,
                                            roww=row_weight if training else None)
            dW1t = _wg_empty(dout, (aggr.size(1), h1.size(1)))
            defer_wgrads([lambda: K.gemm_tn(h1, dout, a_scale=scale, a_shift=shift, out=dW2t), lambda: K.gemm_tn(aggr, dh1, out=dW1t)],
                         (h1, dout, scale, shift, aggr, dh1))
            daggr = K.gemm_nn(dh1, W1, B1n=W1t) if ctx.needs_input_grad[0] else None
            return (daggr, dW1t, None, db1, red[1], red[0], dW2t, None, db2, None, None, None, None, None, None, None, None, None, None)
        dW2t = K.gemm_tn(h1, dout, a_scale=scale, a_shift=shift)
        if oc >= 0:
            db2 = dW2t[oc]
        dr = K.gemm_nn(dout, W2, B1n=W2t)
        red = K.bn_bwd_reduce(dr, h1, mean, invstd, scale, shift)  # [sum dy, sum dy*hhat]
        dbeta, dgamma = red[0], red[1]
        dh1, db1 = K.bn_relu_bwd_colsum(dr, h1, mean, invstd, scale, shift, gamma, red, 1.0 / R if training else 0.0,
                                        roww=row_weight if training else None)
        dW1t = K.gemm_tn(aggr, dh1)
        daggr = K.gemm_nn(dh1, W1, B1n=W1t) if ctx.needs_input_grad[0] else None
        return daggr, dW1t, None, db1, dgamma, dbeta, dW2t, None, db2, None, None, None, None, None, None, None, None, None, None


def _colstats_ok(K, rows, k1, no):
    fn = getattr(K, 'colstats_supported', None)
    return fn is not None and fn(rows, k1, no)


def gat_mlp(aggr, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean, run_var, batch_stats, eps, p, apply_act=True, running=None,
            row_weight=None, ones_col=-1):
    """`batch_stats`: BatchNorm uses batch statistics (train mode); `p`: dropout rate (0 disables); `apply_act`: GELU+dropout
    fused after the second Linear (False returns the raw GATConvE output); `row_weight` [R] (sums to 1): rows enter the batch
    statistics with these weights instead of 1/R (the edge encoder on distinct edge classes, weighted by class counts)."""
    p = float(p) if apply_act else 0.0
    return GatMlpFn.apply(aggr, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean, run_var, batch_stats, eps, p,
                          next_seed() if p > 0 else 0, apply_act, running, row_weight, ones_col)


# ------------------------------------------------------------------------------------------------------------------
# One GATConvE hop as ONE autograd operator over qagnn_hop_{fwd,bwd}_f32 (csrc/hop.hip): the same launches as
# LinearNNFn -> EdgeAttnFn -> GatMlpFn above, sequenced in C.  hop_fwd_composed / hop_bwd_composed are that same sequence
# written against the per-kernel provider interface: the definition the native sequencing is tested against
# (bit-identical), and what a provider without a native hop (the torch emulation in tests/) runs.
#
# Which one runs (QAGNN_FUSED_HOP = 1 / 0 pins it; default "auto"): measured on MI355X (profiles/r1_run58_fused_hop_ab_*.txt),
#   * host-bound batches (the reference's mini-batch of 10 subgraphs: the GPU work of a step is < 2 ms): native sequencing
#     1434 / 1689 vs 1317 / 1373 QA-subgraphs/s;
#   * GPU-bound batches (320 subgraphs, N = 64 000 rows): the composed path 26 098 / 26 113 vs 25 481 / 25 429, because only it
#     can run the weight-gradient GEMMs of one operator under the edge backward of the next (QAGNN_WGRAD_OVERLAP above).
# Rounds 1-5: "auto" took the native hop below 32 768 node rows, where a step is bounded by the host.  Round 6: the natively sequenced stack
# at EVERY size -- it is where the operand maxima of the three-MFMA GEMM form travel from the kernel that produces a tensor to the product
# that reads it (qagnn_hop_args.amax), and the weight-gradient overlap that favoured the composed path is worth 0.5 % since the kernels
# fill the chip on their own (DESIGN.md, round 5 visits 8-9; the native stack forks its weight gradients onto a side stream itself).
_fh = _os.environ.get('QAGNN_FUSED_HOP', 'auto')
FUSED_HOP = {'1': True, '0': False}.get(_fh, None)  # None = auto
FUSED_HOP_MAX_ROWS = None  # (auto: no row limit)
FUSED_STACK = True  # where the native hop is taken, take all k hops in one call (a module attribute: the tests compare the two forms)


def use_fused_hop(n_rows):
    return FUSED_HOP if FUSED_HOP is not None else (FUSED_HOP_MAX_ROWS is None or n_rows < FUSED_HOP_MAX_ROWS)
HOP_PARAMS = ('Wx_t', 'Wx', 'Ws_t', 'Ws', 'TT', 'EkEm', 'W1t', 'W1', 'b1', 'gamma', 'beta', 'W2t', 'W2', 'b2', 'run_mean_p', 'run_var_p')


def hop_fwd_composed(K, graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, running, cols=-1):
    Wx_t, Wx, Ws_t, Ws, TT, EkEm, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean_p, run_var_p = prm
    ones_col = cols[1] if isinstance(cols, tuple) else -1
    KMQ = K.gemm_nn(X, Wx_t, S, Ws_t, rowtab=TT, rowidx=ntype, B1n=Wx, B2n=Ws)
    aggr, a, alpha = K.edge_attn_fwd(graph, KMQ, EkEm, HP, qscale)
    if batch_stats and _colstats_ok(K, aggr.size(0), aggr.size(1), W1t.size(1)):
        h1, part = K.gemm_nn(aggr, W1t, bias=b1, B1n=W1, colstats=True)
        stats = K.bn_stats_finalize(part, aggr.size(0), gamma, beta, eps, running, ones_col)
    else:
        h1 = K.gemm_nn(aggr, W1t, bias=b1, B1n=W1)
        if batch_stats:
            sc = 1.0 / aggr.size(0)
            mean = K.colsum(h1, scale=sc)[0]
            var = K.colvar_sum(h1, mean, scale=sc)
        else:
            mean, var = run_mean_p, run_var_p
        invstd, scale, shift = K.bn_finalize(mean, var, gamma, beta, eps, running, ones_col)
        stats = torch.stack([mean, var, invstd, scale, shift])
    out = K.gemm_nn(h1, W2t, bias=b2, a_scale=stats[3], a_shift=stats[4], B1n=W2)
    y = K.gelu_dropout_fwd(out, p, seed) if apply_act else out
    return y, (KMQ, torch.stack([a, alpha]), aggr, h1, out, stats)


def hop_bwd_composed(K, graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, saved, dy, need_dX, need_dS,
                     dX_acc=None, dS_acc=None, tab_col=-1):
    Wx_t, Wx, Ws_t, Ws, TT, EkEm, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean_p, run_var_p = prm
    KMQ, aa, aggr, h1, out, stats = saved
    mean, invstd, scale, shift = (stats[0] if batch_stats else run_mean_p), stats[2], stats[3], stats[4]
    R = aggr.size(0)
    tab_col, ones_col = tab_col if isinstance(tab_col, tuple) else (tab_col, -1)
    dout = K.gelu_dropout_bwd(out, dy, p, seed) if apply_act else dy
    dW2t = K.gemm_tn(h1, dout, a_scale=scale, a_shift=shift)
    db2 = K.colsum(dout)[0] if ones_col < 0 else dW2t[ones_col]  # (a view: the bias gradient IS that row)
    dr = K.gemm_nn(dout, W2, B1n=W2t)
    red = K.bn_bwd_reduce(dr, h1, mean, invstd, scale, shift)
    dh1, db1 = K.bn_relu_bwd_colsum(dr, h1, mean, invstd, scale, shift, gamma, red, 1.0 / R if batch_stats else 0.0)
    dW1t = K.gemm_tn(aggr, dh1)
    daggr = K.gemm_nn(dh1, W1, B1n=W1t)
    dKMQ, dEkEm = K.edge_attn_bwd(graph, KMQ, EkEm, HP, qscale, aa[0], aa[1], daggr)
    if S is not None:  # one launch for both (qagnn_gemm_tn2_f32), like qagnn_hop_bwd_f32 when its two outputs are adjacent
        joint = K.gemm_tn2(X, S, dKMQ)
        dWx_t, dWs_t = joint[:X.size(1)], joint[X.size(1):]
    else:
        dWx_t, dWs_t = K.gemm_tn(X, dKMQ), None
    if dWs_t is not None and tab_col >= 0:
        dTT = dWs_t[tab_col:tab_col + TT.size(0)]  # (a view) rows of the type indicators in S (qagnn_hop_args.tab_col)
    else:
        dTT = K.colsum(dKMQ, ntype, TT.size(0))
    dX = K.gemm_nn(dKMQ, Wx, out=dX_acc, accumulate=dX_acc is not None, B1n=Wx_t) if need_dX else None
    dS = K.gemm_nn(dKMQ, Ws, out=dS_acc, accumulate=dS_acc is not None, B1n=Ws_t) if (S is not None and need_dS) else None
    return dX, dS, dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, red[1], red[0], dW2t, db2


class HopFn(torch.autograd.Function):
    """(X', a) = one GATConvE hop in packed layout; see qagnn_hop_args in include/qagnn_hip.h for every operand."""

    @staticmethod
    @_fwd
    def forward(ctx, X, S, ntype, graph, HP, qscale, batch_stats, eps, p, seed, apply_act, running, acc, tab_col, *prm):
        K = kernels()
        fwd = getattr(K, 'hop_fwd', None)
        args = (graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act)
        y, saved = fwd(*args, running, tab_col) if fwd is not None else hop_fwd_composed(K, *args, running, tab_col)
        ctx.save_for_backward(X, S, ntype, *prm, *saved)
        ctx.cfg = (graph, HP, qscale, batch_stats, eps, p, seed, apply_act, len(prm))
        ctx.tab_col = tab_col
        ctx.acc = acc if acc is not None else (None, False, None, False)  # GradAcc of X / S and whether this hop returns the totals
        a = saved[1][0]
        ctx.mark_non_differentiable(a)
        ctx.set_materialize_grads(False)  # (no zero-filled [E', 4] stand-in for the gradient of `a`)
        return y, a

    @staticmethod
    @_bwd
    def backward(ctx, dy, _da):
        if dy is None:
            return (None,) * 30
        K = kernels()
        graph, HP, qscale, batch_stats, eps, p, seed, apply_act, nprm = ctx.cfg
        t = ctx.saved_tensors
        X, S, ntype, prm, saved = t[0], t[1], t[2], t[3:3 + nprm], t[3 + nprm:]
        flush_wgrads(dy)  # weight gradients queued by the operators around the stack run on the side stream under this hop
        bwd = getattr(K, 'hop_bwd', None)
        accX, lastX, accS, lastS = ctx.acc
        args = (graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, saved, dy.contiguous(),
                ctx.needs_input_grad[0], ctx.needs_input_grad[1], accX.buf if accX is not None else None,
                accS.buf if accS is not None else None, ctx.tab_col)
        # (overlap: the native hop forks its weight-gradient products onto a side stream of its own, qagnn_hop_args.side_stream)
        dX, dS, dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dgamma, dbeta, dW2t, db2 = (bwd(*args, overlap=WGRAD_OVERLAP) if bwd is not None
                                                                                  else hop_bwd_composed(K, *args))
        if accX is not None and dX is not None:  # dX / dS are the running totals now (accumulated in place when a buffer existed)
            accX.buf, dX = (None, dX) if lastX else (dX, None)
        if accS is not None and dS is not None:
            accS.buf, dS = (None, dS) if lastS else (dS, None)
        #        X   S   ntype graph HP    qscale bstats eps  p     seed  act   running acc  | Wx_t  Wx    Ws_t   Ws    TT   EkEm   W1t   W1   b1
        return (dX, dS, None, None, None, None, None, None, None, None, None, None, None, None, dWx_t, None, dWs_t, None, dTT, dEkEm, dW1t, None,
                db1, dgamma, dbeta, dW2t, None, db2, None, None)


def gat_hop(X, S, ntype, graph, HP, qscale, prm, batch_stats, eps, p, apply_act, running, acc=None, tab_col=-1):
    """prm: the 16 packed operands named in HOP_PARAMS.  `p`: dropout rate after the GELU (0 disables).
    acc = (accX, lastX, accS, lastS): GradAcc of X / S, see GradAcc."""
    p = float(p) if apply_act else 0.0
    return HopFn.apply(X, S, ntype, graph, HP, qscale, batch_stats, eps, p, next_seed() if p > 0 else 0, apply_act, running, acc, tab_col, *prm)


class StackFn(torch.autograd.Function):
    """All k GATConvE hops of QAGNN_Message_Passing.mp_helper (GELU + dropout after each) as ONE autograd node over
    qagnn_stack_{fwd,bwd}_f32: for host-bound batches (the reference's mini-batch of 10 subgraphs) the per-hop Python, ctypes and
    autograd bookkeeping is most of a step.  Same launches, in the same order, as k HopFn nodes."""

    @staticmethod
    @_fwd
    def forward(ctx, X, S, ntype, graph, HP, qscale, batch_stats, eps, p, seeds, runnings, accX, tab_col, k, *prm):
        K = kernels()
        npk = len(prm) // k
        prms = [prm[l * npk:(l + 1) * npk] for l in range(k)]
        xw, sw = amax_lookup(X), amax_lookup(S)
        kw = {}
        if xw is not None or sw is not None:  # (the producers of X / S left their maxima: no reduction pass inside the stack)
            kw = dict(x_amax=xw, s_amax=sw)
        y, saved = K.stack_fwd(graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, runnings, tab_col, **kw)
        if len(saved) > 4 and _wants_amax(K, X) and batch_stats:  # (the library's own condition for the form: csrc/hop.hip, hop_h2)
            amax_note(y, saved[4][k - 1, 4:5])  # max |y| of the last hop (AM_Y), left by its GELU / dropout pass
        ctx.save_for_backward(X, S, ntype, *prm, *saved)
        ctx.cfg = (graph, HP, qscale, batch_stats, eps, p, seeds, k, npk)
        ctx.accX, ctx.tab_col = accX, tab_col
        return y

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        K = kernels()
        graph, HP, qscale, batch_stats, eps, p, seeds, k, npk = ctx.cfg
        t = ctx.saved_tensors
        X, S, ntype, prm, saved = t[0], t[1], t[2], t[3:3 + k * npk], t[3 + k * npk:]
        prms = [prm[l * npk:(l + 1) * npk] for l in range(k)]
        flush_wgrads(dy)  # weight gradients queued by the operators around the stack run on the side stream under the hops
        accX = ctx.accX
        dX, dS, grads = K.stack_bwd(graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, saved, dy.contiguous(),
                                    ctx.needs_input_grad[0], ctx.needs_input_grad[1], accX.buf if accX is not None else None, ctx.tab_col,
                                    overlap=WGRAD_OVERLAP)  # qagnn_hop_args.side_stream: weight gradients beside the data-gradient chain
        if accX is not None:
            accX.buf = None  # handed over: this node is the last reader of the stack input (it was created first)
        out = []
        for (dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dgamma, dbeta, dW2t, db2) in grads:
            #       Wx_t   Wx    Ws_t   Ws    TT   EkEm   W1t   W1   b1   gamma   beta   W2t   W2   b2  run_mean run_var
            out += [dWx_t, None, dWs_t, None, dTT, dEkEm, dW1t, None, db1, dgamma, dbeta, dW2t, None, db2, None, None]
        return (dX, dS, None, None, None, None, None, None, None, None, None, None, None, None, *out)


def gat_stack(X, S, ntype, graph, HP, qscale, prms, batch_stats, eps, p, runnings, accX=None, tab_col=-1):
    """prms: per-layer lists of the 16 packed operands named in HOP_PARAMS."""
    k = len(prms)
    seeds = [next_seed() if p > 0 else 0 for _ in range(k)]
    flat = [t for prm in prms for t in prm]
    return StackFn.apply(X, S, ntype, graph, HP, qscale, batch_stats, eps, float(p), seeds, runnings, accX, tab_col, k, *flat)


class PoolAttnFn(torch.autograd.Function):
    """Node-sized part of the pooling head (utils/layers.py:284-299 inside :344-371): masked softmax attention of NH query
    vectors over the n node rows of every subgraph, attention dropout, weighted sum of the rows.  Returns (z [B, NH, Cc],
    attn_d [B, NH, n] = the attention after dropout, which is what the reference returns as pool_attn)."""

    @staticmethod
    @_fwd
    def forward(ctx, u, cvec, K3, mask, inv_temp, p, seed):
        Kn = kernels()
        u, cvec, K3 = u.contiguous(), cvec.contiguous(), K3.contiguous()
        attn, attn_d, z = Kn.pool_attn_fwd(u, cvec, K3, mask.contiguous(), inv_temp, p, seed)
        ctx.save_for_backward(u, K3, attn, attn_d)
        ctx.cfg = (inv_temp, p, seed)
        ctx.set_materialize_grads(False)
        return z, attn_d

    @staticmethod
    @_bwd
    def backward(ctx, dz, dattn_d):
        u, K3, attn, attn_d = ctx.saved_tensors
        inv_temp, p, seed = ctx.cfg
        if dz is None:
            dz = torch.zeros_like(u)
        dK, du, dc = kernels().pool_attn_bwd(u, K3, inv_temp, p, seed, attn, attn_d, dz.contiguous(),
                                             dattn_d.contiguous() if dattn_d is not None else None)
        return du, dc, dK, None, None, None, None


def pool_attention(u, cvec, K3, mask, inv_temp, p, training):
    p = float(p) if training else 0.0
    return PoolAttnFn.apply(u, cvec, K3, mask, inv_temp, p, next_seed() if p > 0 else 0)


class HeadFn(torch.autograd.Function):
    """Pooling + everything behind it as ONE autograd node (reference utils/layers.py:344-371 and modeling_qagnn.py:178-182 with
    fc_layer_num = 0):  logits[b] = < drop_fc([ drop_pool(Wv pool(K3[b]) + bv sum attn) | sent[b] | K3[b, 0] ]), w_fc > + b_fc.
    Forward = qagnn_pool_attn_fwd_f32 + qagnn_head_post_fwd_f32; backward = qagnn_head_post_bwd_f32, one column sum for
    d w_fc | d bv | d b_fc, one product for d BDv, qagnn_pool_attn_bwd_f32, and the context row's direct gradient added into the
    pooling's dK in place (autograd's slice backward would zero-fill and add a second [B, n, DP] tensor: 53 MB each at 320 subgraphs)."""

    @staticmethod
    @_fwd
    def forward(ctx, u, cvec, K3, mask, inv_temp, p_attn, seed_attn, BDv, bv, sent, w_fc, b_fc, d, p_pool, p_fc, seed_pool, seed_fc):
        Kn = kernels()
        u, cvec, K3, sent = u.contiguous(), cvec.contiguous(), K3.contiguous(), sent.contiguous()
        BDv, bv, w1, b_fc = BDv.contiguous(), bv.contiguous(), w_fc.reshape(-1).contiguous(), b_fc.contiguous()
        attn, attn_d, z = Kn.pool_attn_fwd(u, cvec, K3, mask.contiguous(), inv_temp, p_attn, seed_attn)
        logits, out, asum = Kn.head_post_fwd(z, attn_d, BDv, bv, sent, K3, d, w1, b_fc, p_pool, p_fc, seed_pool, seed_fc)
        ctx.save_for_backward(u, K3, attn, attn_d, z, out, asum, BDv, bv, sent, w1)
        ctx.cfg = (inv_temp, p_attn, seed_attn, d, p_pool, p_fc, seed_pool, seed_fc, w_fc.shape)
        ctx.mark_non_differentiable(attn_d)
        return logits.unsqueeze(1), attn_d

    @staticmethod
    @_bwd
    def backward(ctx, dlogits, _dattn_unused):
        u, K3, attn, attn_d, z, out, asum, BDv, bv, sent, w1 = ctx.saved_tensors
        inv_temp, p_attn, seed_attn, d, p_pool, p_fc, seed_pool, seed_fc, w_shape = ctx.cfg
        Kn = kernels()
        B, NO = out.shape
        Ds, n = sent.size(1), attn.size(2)
        L = NO + Ds + d
        dz, dattn, dout, dsent, dZ, part = Kn.head_post_bwd(dlogits.reshape(-1).contiguous(), out, asum, BDv, bv, sent, K3, d, w1, p_pool, p_fc,
                                                            seed_pool, seed_fc, n, ctx.needs_input_grad[9])
        cols = Kn.colsum(part)[0]
        dBDv = torch.mm(z.reshape(B, -1).t(), dout) if ctx.needs_input_grad[7] else None
        dK, du, dc = Kn.pool_attn_bwd(u, K3, inv_temp, p_attn, seed_attn, attn, attn_d, dz, dattn)
        Kn.add_row0(dK, dZ)
        return (du, dc, dK, None, None, None, None, dBDv, cols[L:L + NO], dsent, cols[:L].reshape(w_shape), cols[L + NO:L + NO + 1],
                None, None, None, None, None)


def head_supported(nh, dv, width, n):
    K = kernels()
    lim = getattr(K, 'HEAD_LIMITS', None)
    # (k_head_post_{fwd,bwd} index the GNN output as H_HEADS = 4 groups of width / 4 floats: the head-padded layout of HeadLayout)
    return (HEAD_FUSED and lim is not None and H_HEADS == 4 and nh <= lim[0] and nh * dv <= lim[1] and width <= lim[2] and width % 4 == 0
            and pool_attention_supported(nh, width, n))


def head(u, cvec, K3, mask, inv_temp, p_attn, BDv, bv, sent, w_fc, b_fc, d, p_pool, p_fc, training):
    """(logits [B, 1], attn_d [B, NH, n]); see HeadFn."""
    p_attn, p_pool, p_fc = (float(p_attn), float(p_pool), float(p_fc)) if training else (0.0, 0.0, 0.0)
    return HeadFn.apply(u, cvec, K3, mask, inv_temp, p_attn, next_seed() if p_attn > 0 else 0, BDv, bv, sent, w_fc, b_fc, int(d),
                        p_pool, p_fc, next_seed() if p_pool > 0 else 0, next_seed() if p_fc > 0 else 0)


def pool_attention_supported(nh, width, n):
    lim = getattr(kernels(), 'POOL_LIMITS', None)
    return lim is not None and nh <= lim[0] and width <= lim[1] and width % 4 == 0 and n <= lim[2]


_TABLE_AMAX = {}


def table_amax(K, table):
    """max |.| of a FROZEN entity table as the device word the three-MFMA GEMM form wants (an upper bound of the maximum over any batch's
    gathered rows): one reduction pass per table and version, cached -- the first call must not fall inside a stream capture (GraphedStep's
    warm-up steps are eager).  None where the form does not apply (table not 16-byte / 4-element aligned, provider without the form)."""
    if not (getattr(K, 'gemm_split', 1) >= 2 and getattr(K, 'name', '') == 'hip') or table.requires_grad or table.numel() % 4 != 0 \
            or table.data_ptr() % 16 != 0 or not table.is_contiguous():
        return None
    key = (table.data_ptr(), table._version, table.numel())
    w = _TABLE_AMAX.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        if len(_TABLE_AMAX) > 8:
            _TABLE_AMAX.clear()
        w = _TABLE_AMAX[key] = K.absmax(table.detach().view(-1))
    return w


class ConceptInputFn(torch.autograd.Function):
    """Node features entering the GNN, fused (reference modeling_qagnn.py:153-156 + utils/layers.py:604-605):

        Hp = dropout_e( gelu( [ ctx_pre[g] ; emb[concept_ids[g, 1:] - 1] @ Wc^T + bc ] ) )        head-padded [N, DP]

    The frozen entity-table gather is folded into the GEMM's operand load (`a_rowidx`), so the [N, concept_in_dim]
    gathered matrix (260 MB at the CSQA batch) is never materialised; `rowidx` is -1 on the context-node rows, whose
    pre-activation `ctx_pre` = svec2nvec(sent_vecs) is written in instead.  The table receives no gradient (frozen,
    freeze_ent_emb=True is the only mode this path takes)."""

    @staticmethod
    @_fwd
    def forward(ctx, emb_w, rowidx, Wc_t, bc, ctx_pre, n, p, seed, Wc=None):
        K = kernels()
        # (the frozen table's maximum: the gathered product then runs in the three-MFMA form like the stack's -- csrc/gemm_nn2.hip)
        tam = table_amax(K, emb_w) if rowidx.numel() >= 8192 else None
        pre = K.gemm_nn(emb_w, Wc_t, bias=bc, a_rowidx=rowidx, B1n=Wc if Wc is not None else Wc_t.t().contiguous(),
                        **(dict(a_amax1=tam) if tam is not None else {}))
        B = ctx_pre.size(0)
        pre.view(B, n, -1)[:, 0] = ctx_pre
        ctx.save_for_backward(emb_w, rowidx, pre)
        ctx.cfg = (n, p, seed)
        if _wants_amax(K, pre):
            Hp, word = K.gelu_dropout_fwd(pre, p, seed, amax=True)
            return amax_note(Hp, word)
        return K.gelu_dropout_fwd(pre, p, seed)

    @staticmethod
    @_bwd
    def backward(ctx, dHp):
        K = kernels()
        emb_w, rowidx, pre = ctx.saved_tensors
        n, p, seed = ctx.cfg
        dpre = K.gelu_dropout_bwd(pre, dHp.contiguous(), p, seed)
        dctx = dpre.view(-1, n, dpre.size(1))[:, 0].contiguous()
        dWc_t, cs = K.gemm_tn(emb_w, dpre, a_rowidx=rowidx), K.colsum(dpre)
        dbc = cs[0] - dctx.sum(0)  # the bias only acts on the entity rows (context-node rows were overwritten)
        return None, None, dWc_t, dbc, dctx, None, None, None, None


def concept_input(emb_w, rowidx, Wc_t, bc, ctx_pre, n, p, training, Wc=None):
    """Wc_t [in, DP], bc [DP]: cpt_transform in the kernels' layout; Wc [DP, in]: the same weight the other way round (optional: saves
    a transpose copy per step when the caller packs both with one gather)."""
    p = float(p) if training else 0.0
    return ConceptInputFn.apply(emb_w, rowidx, Wc_t, bc, ctx_pre, n, p, next_seed() if p > 0 else 0, Wc)


QSCALE = lambda dh: 1.0 / math.sqrt(dh)  # noqa: E731  (query / sqrt(dim_per_head), modeling_qagnn.py:469)
