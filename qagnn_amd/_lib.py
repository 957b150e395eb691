"""ctypes binding of libqagnn_hip.so (include/qagnn_hip.h) + a thin tensor-level wrapper.

PyTorch is used here for exactly three things: device memory (torch.empty), the current HIP stream handle and
raw data pointers.  All compute goes through the C ABI.  There is no fallback: if the shared library is missing
or no MI355X is visible, `HipKernels()` raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QAGNN_LIB') or os.path.join(_HERE, 'libqagnn_hip.so')  # QAGNN_LIB: an alternate build (kernel A/B runs)

EXPORTS = ['qagnn_last_error', 'qagnn_abi_version', 'qagnn_graph_storage_elems', 'qagnn_graph_prep', 'qagnn_graph_prep_blocked',
           'qagnn_graph_from_blobs', 'qagnn_radam_step_f32', 'qagnn_node_prep_f32', 'qagnn_seed_epoch_advance', 'qagnn_seed_epoch_set',
           'qagnn_gemm_nn_f32', 'qagnn_gemm_nn_split_f32', 'qagnn_gemm_nn_pack_bytes', 'qagnn_gemm_nn_ws_bytes', 'qagnn_gemm_nn_split_ws_f32', 'qagnn_gemm_nn_prepack_bytes', 'qagnn_gemm_nn_prepack_f32',
           'qagnn_gemm_nn_prepack_clear', 'qagnn_gemm_tn_workspace_elems', 'qagnn_gemm_tn_f32', 'qagnn_gemm_tn2_f32', 'qagnn_gemm_tn_colsum_f32',
           'qagnn_colreduce_workspace_elems', 'qagnn_colreduce_f32', 'qagnn_bn_finalize_f32', 'qagnn_bn_stats_finalize_f32', 'qagnn_bn_relu_bwd_f32',
           'qagnn_gelu_dropout_fwd_f32', 'qagnn_gelu_dropout_bwd_f32', 'qagnn_sin_basis_f32',
           'qagnn_bn_relu_bwd_colsum_f32',
           'qagnn_pool_attn_fwd_f32', 'qagnn_pool_attn_bwd_f32', 'qagnn_head_post_fwd_f32', 'qagnn_head_post_bwd_f32', 'qagnn_add_row0_f32', 'qagnn_gather_multi_f32', 'qagnn_gather_multi_sum_f32',
           'qagnn_edge_attn_fwd_f32', 'qagnn_edge_attn_bwd_f32',
           'qagnn_hop_fwd_workspace_elems', 'qagnn_hop_bwd_workspace_elems', 'qagnn_hop_fwd_f32', 'qagnn_hop_bwd_f32',
           'qagnn_stack_fwd_f32', 'qagnn_stack_bwd_f32', 'qagnn_absmax_f32', 'qagnn_zero_words', 'qagnn_gemm_tn_h2_f32', 'qagnn_gelu_dropout_fwd_amax_f32', 'qagnn_gelu_dropout_amax_scratch_elems',
           'qagnn_timing_enable', 'qagnn_timing_read', 'qagnn_gemm_tn_h1_f32', 'qagnn_gelu_dropout_bwd_amax_f32']

CLS_SLICES = 4  # QAGNN_CLS_SLICES
ABI_VERSION = 21  # bumped when the ABI of include/qagnn_hip.h changes (2: qagnn_graph.tgt_t; 3: (group, class) class order; 4: qagnn_hop_args.accumulate_dX; 5: qagnn_graph_from_blobs; 6: qagnn_edge_attn_fwd_lds_f32 replaces the first LDS kernel; 7: qagnn_graph.pk_s / pk_t / sub_ncls / sub_cls; 8: qagnn_hop_args.gemm_split; 9: qagnn_stack_{fwd,bwd}_f32; 10: qagnn_node_prep_f32 checks the concept ids; 11: qagnn_graph_from_blobs takes an edge CAPACITY, seed epoch, column statistics in the GEMM epilogue, LDS-resident edge forward removed; 12: qagnn_hop_args.side_stream, two buffer sets in the backward workspace; 13: qagnn_gemm_tn2_f32; 14: qagnn_gemm_nn_split_ws_f32 / qagnn_gemm_nn_pack_bytes, hop workspaces carry the pack buffer, qagnn_gemm_nn_prepack_{bytes,f32,clear}; 15: qagnn_head_post_{fwd,bwd}_f32, qagnn_add_row0_f32, qagnn_gather_multi{,_sum}_f32; 16: qagnn_gemm_nn_ws_bytes; 17: the three-MFMA GEMM form -- qagnn_gemm_nn_args.a_amax1 / a_amax2, qagnn_pack_desc.pieces, qagnn_hop_args.amax, qagnn_absmax_f32, qagnn_zero_words, qagnn_gemm_tn_h2_f32; 18: qagnn_hop_args.x_amax / s_amax, qagnn_gelu_dropout_fwd_amax_f32, qagnn_timing_enable / qagnn_timing_read; 19: the reduced-precision form on request -- qagnn_gemm_nn_args.pieces, qagnn_gemm_tn_h1_f32, qagnn_hop_args.gemm_split == 3; 20: qagnn_gelu_dropout_bwd_amax_f32; 21: qagnn_gemm_nn_args.a_rows)

_i32, _i64, _f32, _u64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_void_p


class qagnn_graph(C.Structure):
    _fields_ = ([(n, _i32) for n in ('N', 'E', 'Ep', 'R', 'T', 'C')] +
                [(n, _vp) for n in ('rowptr_s', 'tgt_s', 'src_s', 'cls_s', 'eid_s', 'rowptr_t', 'src_t', 'tgt_t', 'cls_t', 'pos_t',
                                    'cls_count', 'src_c', 'tgt_c', 'pos_c', 'chunk_cls', 'chunk_beg',
                                    'chunk_len', 'n_chunks', 'chunkptr')] +
                [('max_chunks', _i32), ('err', _vp), ('block_n', _i32), ('n_groups', _i32)])


GATHER_MAX = 160


class qagnn_gather_tabs(C.Structure):
    _fields_ = [('p', _vp * GATHER_MAX), ('n', _i32)]


class qagnn_pack_desc(C.Structure):
    _fields_ = [('B1n', _vp), ('ldn1', _i32), ('K1', _i32), ('B2n', _vp), ('ldn2', _i32), ('K2', _i32), ('No', _i32), ('pieces', _i32)]


class qagnn_gemm_nn_args(C.Structure):
    _fields_ = [('A1', _vp), ('lda1', _i32), ('K1', _i32), ('B1', _vp), ('ldb1', _i32),
                ('A2', _vp), ('lda2', _i32), ('K2', _i32), ('B2', _vp), ('ldb2', _i32),
                ('C', _vp), ('ldc', _i32), ('M', _i32), ('No', _i32),
                ('bias', _vp), ('rowtab', _vp), ('ldt', _i32), ('rowidx', _vp),
                ('a_scale', _vp), ('a_shift', _vp), ('accumulate', _i32), ('a_rowidx', _vp), ('xcd_remap', _i32), ('colstat_part', _vp),
                ('a_amax1', _vp), ('a_amax2', _vp), ('a_rows', _i64), ('pieces', _i32)]


class qagnn_hop_args(C.Structure):
    """Mirror of qagnn_hop_args in include/qagnn_hip.h (field order is the ABI)."""
    _fields_ = ([('g', C.POINTER(qagnn_graph))] + [(n, _i32) for n in ('N', 'DP', 'SP', 'HP', 'T')] + [('qscale', _f32)] +
                [(n, _vp) for n in ('X', 'S', 'ntype', 'Wx_t', 'Wx', 'Ws_t', 'Ws', 'TT', 'EkEm', 'W1t', 'W1', 'b1', 'gamma', 'beta',
                                    'W2t', 'W2', 'b2')] +
                [('batch_stats', _i32), ('eps', _f32), ('run_mean_p', _vp), ('run_var_p', _vp), ('run_mean', _vp), ('run_var', _vp),
                 ('num_batches_tracked', _vp), ('dense_pos', _vp), ('d', _i32), ('momentum', _f32),
                 ('apply_act', _i32), ('p_drop', _f32), ('seed', _u64)] +
                [(n, _vp) for n in ('KMQ', 'a', 'alpha', 'aggr', 'h1', 'out', 'y', 'stats', 'dy', 'dX', 'dS')] +
                [('accumulate_dS', _i32), ('accumulate_dX', _i32)] +
                [(n, _vp) for n in ('dWx_t', 'dWs_t', 'dTT', 'dEkEm', 'dW1t', 'db1', 'dbn', 'dW2t', 'db2', 'ws')] +
                [('ws_elems', _i64), ('gemm_split', _i32), ('ones_col', _i32), ('tab_col', _i32), ('side_stream', _vp), ('amax', _vp), ('x_amax', _vp), ('s_amax', _vp)])


HOP_AMAX_WORDS = 16  # QAGNN_HOP_AMAX_WORDS


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library and declare prototypes.  Raises OSError if it has not been built."""
    if not os.path.exists(path):
        raise OSError(f'{path} not found: build it with `python -m qagnn_amd.build` (hipcc, gfx950)')
    lib = C.CDLL(path)
    lib.qagnn_last_error.restype = C.c_char_p
    lib.qagnn_abi_version.restype = _i32
    lib.qagnn_graph_storage_elems.restype = _i64
    lib.qagnn_graph_storage_elems.argtypes = [_i32, _i32, _i32, _i32]
    lib.qagnn_graph_prep.argtypes = [C.POINTER(qagnn_graph), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]
    lib.qagnn_graph_prep_blocked.argtypes = [C.POINTER(qagnn_graph), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]
    lib.qagnn_graph_from_blobs.argtypes = [C.POINTER(qagnn_graph), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]
    lib.qagnn_node_prep_f32.argtypes = [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
    lib.qagnn_seed_epoch_advance.argtypes = [_u64, _vp]
    lib.qagnn_seed_epoch_set.argtypes = [_u64, _vp]
    lib.qagnn_radam_step_f32.argtypes = [_i32, _vp, _vp, _vp, _vp, _vp] + [C.c_double] * 6 + [_i32, _vp]
    lib.qagnn_gemm_nn_f32.argtypes = [C.POINTER(qagnn_gemm_nn_args), _vp]
    lib.qagnn_gemm_nn_split_f32.argtypes = [C.POINTER(qagnn_gemm_nn_args), _vp, _i32, _vp, _i32, _vp]
    lib.qagnn_gemm_nn_pack_bytes.restype = _i64
    lib.qagnn_gemm_nn_pack_bytes.argtypes = [_i32, _i32, _i32]
    lib.qagnn_gemm_nn_split_ws_f32.argtypes = [C.POINTER(qagnn_gemm_nn_args), _vp, _i32, _vp, _i32, _vp, _i64, _vp]
    lib.qagnn_gemm_nn_ws_bytes.restype = _i64
    lib.qagnn_gemm_nn_ws_bytes.argtypes = [C.POINTER(qagnn_gemm_nn_args), _vp, _i32, _vp, _i32]
    lib.qagnn_gemm_nn_prepack_bytes.restype = _i64
    lib.qagnn_gemm_nn_prepack_bytes.argtypes = [C.POINTER(qagnn_pack_desc), _i32]
    lib.qagnn_gemm_nn_prepack_f32.argtypes = [C.POINTER(qagnn_pack_desc), _i32, _vp, _i64, _i64, _vp]
    lib.qagnn_gemm_nn_prepack_clear.argtypes = [_i64]
    lib.qagnn_gemm_tn_workspace_elems.restype = _i64
    lib.qagnn_gemm_tn_workspace_elems.argtypes = [_i32, _i32, _i32]
    lib.qagnn_gemm_tn_f32.argtypes = [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp]
    lib.qagnn_gemm_tn2_f32.argtypes = [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp]
    lib.qagnn_gemm_tn_h2_f32.argtypes = [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.qagnn_gemm_tn_h1_f32.argtypes = lib.qagnn_gemm_tn_h2_f32.argtypes
    lib.qagnn_absmax_f32.argtypes = [_vp, _i64, _vp, _vp]
    lib.qagnn_zero_words.argtypes = [_vp, _i64, _vp]
    lib.qagnn_timing_enable.argtypes = [_i32]
    lib.qagnn_timing_read.argtypes = [_vp, _vp]
    lib.qagnn_gemm_tn_colsum_f32.argtypes = [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                             _vp, _vp]
    lib.qagnn_colreduce_workspace_elems.restype = _i64
    lib.qagnn_colreduce_workspace_elems.argtypes = [_i32, _i32, _i32]
    lib.qagnn_colreduce_f32.argtypes = [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp]
    lib.qagnn_bn_finalize_f32.argtypes = [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp]
    lib.qagnn_bn_stats_finalize_f32.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp]
    lib.qagnn_bn_relu_bwd_f32.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp]
    lib.qagnn_gelu_dropout_fwd_f32.argtypes = [_vp, _vp, _i64, _f32, _u64, _vp]
    lib.qagnn_gelu_dropout_bwd_f32.argtypes = [_vp, _vp, _vp, _i64, _f32, _u64, _vp]
    lib.qagnn_gelu_dropout_fwd_amax_f32.argtypes = [_vp, _vp, _i64, _f32, _u64, _vp, _vp, _vp]
    lib.qagnn_gelu_dropout_bwd_amax_f32.argtypes = [_vp, _vp, _vp, _i64, _f32, _u64, _vp, _vp, _vp]
    lib.qagnn_gelu_dropout_amax_scratch_elems.restype = _i64
    lib.qagnn_gelu_dropout_amax_scratch_elems.argtypes = [_i64]
    lib.qagnn_sin_basis_f32.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _vp]
    lib.qagnn_bn_relu_bwd_colsum_f32.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp]
    lib.qagnn_pool_attn_fwd_f32.argtypes = [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _vp, _vp, _vp, _vp]
    lib.qagnn_pool_attn_bwd_f32.argtypes = [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]
    lib.qagnn_head_post_fwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _u64,
                                            _vp, _vp, _vp, _vp]
    lib.qagnn_head_post_bwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _u64,
                                            _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]
    lib.qagnn_add_row0_f32.argtypes = [_vp, _i64, _vp, _i32, _i32, _vp]
    lib.qagnn_gather_multi_f32.argtypes = [C.POINTER(qagnn_gather_tabs), _vp, _vp, _vp, _i32, _vp]
    lib.qagnn_gather_multi_sum_f32.argtypes = [C.POINTER(qagnn_gather_tabs), _vp, _vp, _i32, _i32, _vp, _vp]
    lib.qagnn_edge_attn_fwd_f32.argtypes = [C.POINTER(qagnn_graph), _vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp]
    lib.qagnn_edge_attn_bwd_f32.argtypes = [C.POINTER(qagnn_graph), _vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _i32,
                                            _vp, _vp, _vp, _vp, _vp, _vp]
    lib.qagnn_hop_fwd_workspace_elems.restype = _i64
    lib.qagnn_hop_fwd_workspace_elems.argtypes = [_i32, _i32, _i32]
    lib.qagnn_hop_bwd_workspace_elems.restype = _i64
    lib.qagnn_hop_bwd_workspace_elems.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.qagnn_hop_fwd_f32.argtypes = [C.POINTER(qagnn_hop_args), _vp]
    lib.qagnn_hop_bwd_f32.argtypes = [C.POINTER(qagnn_hop_args), _vp]
    lib.qagnn_stack_fwd_f32.argtypes = [C.POINTER(qagnn_hop_args), _i32, _vp]
    lib.qagnn_stack_bwd_f32.argtypes = [C.POINTER(qagnn_hop_args), _i32, _vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ('qagnn_abi_version',):
            fn.restype = _i32
    return lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _chk2d(t, name, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.is_contiguous(), \
        f'{name}: need a contiguous 2-D {dtype} device tensor, got {tuple(t.shape)} {t.dtype} {t.device}'
    return t


VALIDATE = os.environ.get('QAGNN_VALIDATE', '0') == '1'  # synchronous input validation (debugging corrupt batches)


class _ErrWatch:
    """Surfaces the device-side input-validation flag of qagnn_graph_prep (err[0]: an edge endpoint, relation id or node type
    was out of range and has been clamped) without a host synchronisation on the hot path.

    The reference raises on such input in its one-hot / index ops (modeling_qagnn.py:352-367, 419-433).  Here the four flag words
    are copied to pinned host memory right behind the preparation kernels, and looked at when the NEXT batch is prepared (or at
    `poll(block=True)`, e.g. at the end of an epoch): a corrupt batch raises one step late instead of training silently on a
    clamped graph.  QAGNN_VALIDATE=1 checks synchronously, in the call that saw the bad batch."""

    def __init__(self):
        self.pending = []
        self.sink = None  # while a hipGraph is being captured (graphed.GraphedStep): the flag tensors the captured launches write

    def watch(self, flags, what, reset=None):
        """`reset`: a persistent flag tensor to clear once its error has been reported (per-call flag arrays need none)."""
        if torch.cuda.is_current_stream_capturing():
            # no copy / event inside a capture; the capturing step keeps the (static) flag tensors and hands them to after_replay()
            # behind every replay, so a bad batch raises one step late under replay exactly as it does on the eager path
            if self.sink is not None:
                self.sink.append((flags, what, reset))
            return
        host = torch.empty(4, dtype=torch.int32, pin_memory=True)
        host.copy_(flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, host, what, reset))
        if VALIDATE:
            self.poll(block=True)

    def after_replay(self, watched):
        """Behind graph.replay(): the flag words the replayed launches wrote -> pinned host memory (async) + an event, as watch()."""
        for flags, what, reset in watched:
            host = torch.empty(4, dtype=torch.int32, pin_memory=True)
            host.copy_(flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((ev, host, what, reset))
        if VALIDATE:
            self.poll(block=True)

    def poll(self, block=False):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return  # event queries are not legal while a hipGraph is being captured; the flags are looked at by the next eager call
        keep, bad = [], None
        for ev, host, what, reset in self.pending:
            if block:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, what, reset))
            elif int(host[0]) != 0:
                if reset is not None:
                    reset.zero_()
                if bad is None:
                    bad = what
        self.pending = keep
        if bad is not None:
            raise RuntimeError(f'out-of-range input in {bad}: the device replaced it by a safe value; the results of that batch '
                               'are not the reference\'s (which raises in its one-hot / index ops)')


ERR_WATCH = _ErrWatch()


class HipGraph:
    """Device-side prepared graph (see qagnn_graph in include/qagnn_hip.h)."""

    def __init__(self, storage, cstruct, N, E, R, T, block_n=0):
        self.storage, self.c = storage, cstruct
        self.N, self.E, self.Ep, self.R, self.T = N, E, E + N, R, T
        self.block_n = block_n
        self.dynamic = False
        self.C = R * T * T + T
        self.max_chunks = cstruct.max_chunks

    def array(self, name, length):
        """int32 view of one of the struct's arrays (for tests and attention-weight export)."""
        off = (getattr(self.c, name) - self.storage.data_ptr()) // 4
        return self.storage[off:off + length]

    @property
    def cls_count(self):
        return self.array('cls_count', self.C)

    @property
    def eid_s(self):
        return self.array('eid_s', self.Ep)


def _device_of(args, kwargs):
    """Device of the first device tensor (or prepared graph) among the arguments of a kernel call."""
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a.device
        elif isinstance(a, HipGraph):
            return a.storage.device
        elif isinstance(a, (tuple, list)):
            for t in a:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    return t.device
    return None


def _on_operand_device(fn):
    """Run a kernel call with the operands' device current, on THAT device's current stream.

    The reference places the LM encoder on cuda:0 and the decoder on cuda:1 whenever two GPUs are visible
    (reference qagnn.py:133-134, 168-169), so the process's current device is not, in general, the device that holds the
    decoder's tensors.  A kernel launched on device 0's stream with device-1 pointers would fault or race with the torch ops
    queued on cuda:1's stream; every entry point of the provider therefore switches to the operands' device first (a no-op
    costing one integer compare when it already is the current one)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = _device_of(args, kwargs)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapped


class _GuardedMeta(type):
    def __new__(mcls, name, bases, ns):
        for k, v in list(ns.items()):
            if callable(v) and not k.startswith('_'):
                ns[k] = _on_operand_device(v)
        return super().__new__(mcls, name, bases, ns)


class HipKernels(metaclass=_GuardedMeta):
    """Tensor-level calls into libqagnn_hip.so, each on the current HIP stream of the device that holds its operands
    (see _on_operand_device)."""
    name = 'hip'

    def __init__(self):
        if not torch.cuda.is_available():
            raise RuntimeError('qagnn_amd needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback')
        self.lib = load_library()
        if self.lib.qagnn_abi_version() != ABI_VERSION:
            raise RuntimeError('libqagnn_hip.so ABI version mismatch')
        # NN GEMMs on the bf16 matrix cores by exact 3-way operand splitting (qagnn_gemm_nn_split_f32) whenever the caller also
        # hands over B in its [No, K] layout; QAGNN_GEMM_SPLIT=0 pins the fp32-MFMA kernels
        # 2 (default): additionally the THREE-MFMA form (scaled two-piece fp16 split, csrc/gemm_nn2.hip) wherever the operand maxima are
        # known -- inside the natively sequenced hops; 1 pins the exact 3 x bf16 split everywhere; 3 asks for the REDUCED-PRECISION form
        # (ONE fp16 MFMA per product where 2 takes three: the GEMM arithmetic of the reference under its --fp16 autocast; never a default)
        self.gemm_split = {'0': 0, '1': 1, '3': 3}.get(os.environ.get('QAGNN_GEMM_SPLIT', '2'), 2)
        self.PACK_MIN_M = 8192  # (the library applies the same threshold: nn2_packed_ok)
        self._side_streams = {}  # per device: the stream the natively sequenced hops put their weight-gradient products on

    # -- helpers -----------------------------------------------------------------------------------------------
    def _stream(self):
        # raw handle of the current device's current stream (torch.cuda.current_stream() builds a Stream object: ~10 us per call,
        # 25 calls per step); the entry points run under _on_operand_device, so "current device" is the operands' device
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())

    def _side_stream(self):
        dev = torch.cuda.current_device()
        st = self._side_streams.get(dev)
        if st is None:
            st = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        return st.cuda_stream

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed (code {rc}): {self.lib.qagnn_last_error().decode()}')

    # -- graph -------------------------------------------------------------------------------------------------
    def graph_prep(self, edge_index, edge_type, node_type, n_etype, n_ntype, block_n=0):
        """block_n = n > 0: the caller expects subgraph i to own node rows [i*n, (i+1)*n) (LM_QAGNN.batch_graph); whether the
        edges really respect that is recorded in a device flag and decides, on the device, which edge kernel runs."""
        assert edge_index.dtype == torch.long and edge_type.dtype == torch.long and node_type.dtype == torch.long
        assert edge_index.dim() == 2 and edge_index.size(0) == 2
        edge_index, edge_type, node_type = edge_index.contiguous(), edge_type.contiguous(), node_type.contiguous()
        N, E = node_type.numel(), edge_index.size(1)
        elems = self.lib.qagnn_graph_storage_elems(N, E, n_etype, n_ntype)
        storage = torch.empty(elems, dtype=torch.int32, device=node_type.device)
        g = qagnn_graph()
        if block_n and N % block_n:
            block_n = 0
        rc = self.lib.qagnn_graph_prep_blocked(C.byref(g), storage.data_ptr(), _ptr(edge_index) if E else None,
                                               _ptr(edge_type) if E else None, node_type.data_ptr(), N, E, n_etype, n_ntype,
                                               int(block_n), self._stream())
        self._check(rc, 'qagnn_graph_prep_blocked')
        G = HipGraph(storage, g, N, E, n_etype, n_ntype, int(block_n))
        ERR_WATCH.poll()  # flags of earlier batches that have landed since
        ERR_WATCH.watch(G.array('err', 4), f'the graph of the batch with N={N} node rows, E={E} edges (edge endpoint / relation id / node type)')
        return G

    def graph_from_blobs(self, packed, node_type):
        """packed: data_utils.PackedGraphBatch on the device (the batch's load-time blobs); node_type [B*n] int64.
        packed.e_cap (optional, >= packed.E): lay the arrays out for that many edges -- every launch shape of the step then depends
        on (B, n, e_cap) only and the true count is read on the device, which is what lets one captured hipGraph serve all batches
        of a capacity bucket (qagnn_amd.graphed)."""
        assert packed.buf.is_cuda and packed.buf.dtype == torch.int32 and node_type.dtype == torch.long and node_type.is_contiguous()
        B, n, R, T = packed.B, packed.n, packed.n_etype, packed.n_ntype
        e_cap = getattr(packed, 'e_cap', None)
        E = packed.E if e_cap is None else int(e_cap)
        assert E >= packed.E, f'edge capacity {E} below the batch\'s {packed.E} edges'
        N = B * n
        assert node_type.numel() == N
        elems = self.lib.qagnn_graph_storage_elems(N, E, R, T)
        storage = torch.empty(elems, dtype=torch.int32, device=node_type.device)
        g = qagnn_graph()
        base = packed.buf.data_ptr()
        rc = self.lib.qagnn_graph_from_blobs(C.byref(g), storage.data_ptr(), base + 4 * packed.head, base, base + 4 * (B + 1),
                                             node_type.data_ptr(), B, n, E, R, T, self._stream())
        self._check(rc, 'qagnn_graph_from_blobs')
        G = HipGraph(storage, g, N, E, R, T, n)
        G.dynamic = e_cap is not None  # E / Ep are capacities: the true E' lives on the device (rowptr_s[N] = sum of cls_count)
        G.keep = packed.buf  # the blobs are read by the kernel just enqueued
        ERR_WATCH.poll()
        ERR_WATCH.watch(G.array('err', 4), f'the graph of the blob batch with B={B} samples, E={packed.E} edges (edge endpoint / relation id / node type)')
        return G

    def seed_epoch_advance(self, delta=1):
        """Advance this device's dropout seed epoch (see qagnn_seed_epoch_advance): the last launch of a captured training step."""
        self._check(self.lib.qagnn_seed_epoch_advance(int(delta), self._stream()), 'qagnn_seed_epoch_advance')

    def seed_epoch_set(self, value=0):
        self._check(self.lib.qagnn_seed_epoch_set(int(value), self._stream()), 'qagnn_seed_epoch_set')

    def node_prep(self, node_scores, adj_lengths, node_type_ids, concept_ids, table_rows=0):
        """-> (normalised scores [B, n] fp32, pooling mask [B, n] bool, entity-table row ids [B*n] int64); qagnn_node_prep_f32.
        table_rows > 0: concept ids outside the entity table become the zero row and are reported through ERR_WATCH (the
        reference's nn.Embedding raises on them)."""
        B, n = node_type_ids.shape
        raw = node_scores.reshape(B, n).float().contiguous()  # the reference computes the normalisation in fp32 (qagnn.py: fp32 inputs)
        assert raw.dtype == torch.float32 and raw.is_contiguous() and adj_lengths.dtype == torch.long and adj_lengths.is_contiguous()
        assert node_type_ids.dtype == torch.long and node_type_ids.is_contiguous() and concept_ids.dtype == torch.long and concept_ids.is_contiguous()
        dev = node_type_ids.device
        score = torch.empty((B, n), dtype=torch.float32, device=dev)
        mask = torch.empty((B, n), dtype=torch.bool, device=dev)
        ridx = torch.empty(B * n, dtype=torch.long, device=dev)
        # a fresh flag per call, like graph_prep's err words: a persistent one would still read 1 in the batch AFTER a bad id
        flags = torch.zeros(4, dtype=torch.int32, device=dev) if table_rows > 0 else None
        rc = self.lib.qagnn_node_prep_f32(raw.data_ptr(), adj_lengths.data_ptr(), node_type_ids.data_ptr(), concept_ids.data_ptr(), B, n,
                                          score.data_ptr(), mask.data_ptr(), ridx.data_ptr(), int(table_rows), _ptr(flags), self._stream())
        self._check(rc, 'qagnn_node_prep_f32')
        if flags is not None:
            ERR_WATCH.poll()
            ERR_WATCH.watch(flags, 'concept_ids (an id outside the entity table)')
        return score, mask, ridx

    def radam_step(self, params, grads, exp_avgs, exp_avg_sqs, beta1, beta2, eps, lr, weight_decay, step_size, mode):
        """One fused RAdam update of a list of fp32 device tensors that share a step count (qagnn_radam_step_f32)."""
        n = len(params)
        for group in (params, grads, exp_avgs, exp_avg_sqs):
            assert len(group) == n and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in group)
        assert all(p.numel() == g.numel() == m.numel() == v.numel() for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs))
        tab = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        numel = (C.c_int64 * n)(*[t.numel() for t in params])
        rc = self.lib.qagnn_radam_step_f32(n, tab(params), tab(grads), tab(exp_avgs), tab(exp_avg_sqs), numel, float(beta1), float(beta2),
                                           float(eps), float(lr), float(weight_decay), float(step_size), int(mode), self._stream())
        self._check(rc, 'qagnn_radam_step_f32')

    # -- GEMMs ---------------------------------------------------------------------------------------------------
    STAT_TILE = 128  # rows per tile of the column statistics a GEMM can leave behind (gemm_split.hip: SBM)

    def colstats_supported(self, M, K1, No):
        """Can gemm_nn(..., colstats=True) deliver per-tile column statistics for this shape?  (the bf16-split kernel, 193..208 columns)"""
        return self.gemm_split and 192 < No <= 208 and K1 % 4 == 0 and M * max(K1, No) * 4 < 2 ** 31 - 1

    def prepack(self, pairs, tag):
        """Pack the B operands of the coming large NN products in ONE launch (qagnn_gemm_nn_prepack_f32) and register them under `tag`:
        pairs = [(B1n, B2n or None), ...], each weight in its [No, K] layout exactly as gemm_nn() will receive it.  Returns what must
        stay alive (and unchanged) until the tag is cleared or packed again: the packed buffer and the weights themselves."""
        descs = (qagnn_pack_desc * len(pairs))()
        for d, pr in zip(descs, pairs):
            b1, b2 = pr[0], pr[1]
            d.pieces = pr[2] if len(pr) > 2 else 3  # (b1, b2, 2): the two scaled fp16 images of the three-MFMA form
            _chk2d(b1, 'B1n')
            d.B1n, d.ldn1, d.K1, d.No = b1.data_ptr(), b1.size(1), b1.size(1), b1.size(0)
            if b2 is not None:
                _chk2d(b2, 'B2n')
                assert b2.size(0) == b1.size(0)
                d.B2n, d.ldn2, d.K2 = b2.data_ptr(), b2.size(1), b2.size(1)
        nbytes = self.lib.qagnn_gemm_nn_prepack_bytes(descs, len(pairs))
        out = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=pairs[0][0].device)
        self._check(self.lib.qagnn_gemm_nn_prepack_f32(descs, len(pairs), out.data_ptr(), out.numel(), int(tag), self._stream()),
                    'qagnn_gemm_nn_prepack_f32')
        # (detached aliases: they pin the storage, not the autograd graph that produced the weights -- a kept-alive graph of the previous
        # iteration makes its AccumulateGrad nodes run on THEIR stream during a later hipGraph capture, which breaks the capture)
        return out, [t.detach() for pr in pairs for t in pr[:2] if t is not None]

    def prepack_clear(self, tag=0):
        self.lib.qagnn_gemm_nn_prepack_clear(int(tag))

    def gemm_nn(self, A1, B1, A2=None, B2=None, bias=None, rowtab=None, rowidx=None, a_scale=None, a_shift=None,
                out=None, accumulate=False, a_rowidx=None, B1n=None, B2n=None, colstats=False, a_amax1=None, a_amax2=None):
        """B1n / B2n: the same weights as B1 / B2 in their [No, K] layout (optional); with them the product runs on the bf16 matrix
        cores by exact operand splitting (see gemm_split.hip), else on the fp32-input MFMAs.
        colstats=True (only where colstats_supported()): returns (C, part) with part [ceil(M/128), 3, No] = per 128-row tile x0 | S1 | S2
        of C's columns, for bn_stats_finalize."""
        _chk2d(A1, 'A1'), _chk2d(B1, 'B1')
        K1 = A1.size(1)
        M = A1.size(0) if a_rowidx is None else a_rowidx.numel()
        No = B1.size(1)
        assert B1.size(0) == K1
        a = qagnn_gemm_nn_args()
        a.A1, a.lda1, a.K1, a.B1, a.ldb1 = A1.data_ptr(), K1, K1, B1.data_ptr(), No
        if A2 is not None:
            _chk2d(A2, 'A2'), _chk2d(B2, 'B2')
            assert A2.size(0) == M and B2.shape == (A2.size(1), No)
            a.A2, a.lda2, a.K2, a.B2, a.ldb2 = A2.data_ptr(), A2.size(1), A2.size(1), B2.data_ptr(), No
        if out is None:
            assert not accumulate
            out = torch.empty((M, No), dtype=torch.float32, device=A1.device)
        else:
            _chk2d(out, 'out')
            assert out.shape == (M, No)
        a.C, a.ldc, a.M, a.No = out.data_ptr(), No, M, No
        if bias is not None:
            assert bias.is_contiguous() and bias.numel() == No and bias.dtype == torch.float32
            a.bias = bias.data_ptr()
        if rowtab is not None:
            _chk2d(rowtab, 'rowtab')
            assert rowtab.size(1) == No and rowidx.dtype == torch.long and rowidx.numel() == M and rowidx.is_contiguous()
            a.rowtab, a.ldt, a.rowidx = rowtab.data_ptr(), No, rowidx.data_ptr()
        if a_scale is not None:
            assert a_scale.numel() == K1 and a_shift.numel() == K1 and a_scale.is_contiguous() and a_shift.is_contiguous()
            a.a_scale, a.a_shift = a_scale.data_ptr(), a_shift.data_ptr()
        a.accumulate = 1 if accumulate else 0
        if a_amax1 is not None and self.gemm_split >= 2:  # int32 [1] device words: absmax() of A1 / A2 -> the three-MFMA form
            assert a_amax1.dtype == torch.int32 and a_amax1.is_cuda and (A2 is None or a_amax2 is not None)
            a.a_amax1, a.a_amax2 = a_amax1.data_ptr(), _ptr(a_amax2)
            a.pieces = 1 if self.gemm_split == 3 else 0  # (3: the reduced-precision form, on request only)
        if a_rowidx is not None:
            assert a_rowidx.dtype == torch.long and a_rowidx.is_contiguous() and a_rowidx.is_cuda
            a.a_rowidx = a_rowidx.data_ptr()
            a.a_rows = A1.size(0)  # (the table's extent: lets the gathered product take the second-generation kernels)
        part = None
        if colstats:
            assert self.colstats_supported(M, K1, No) and B1n is not None and A2 is None and rowtab is None and a_scale is None and a_rowidx is None \
                and not accumulate, 'gemm_nn(colstats=True): bias-only epilogue on the split kernel'
            part = torch.empty((-(-M // self.STAT_TILE), 3, No), dtype=torch.float32, device=A1.device)
            a.colstat_part = part.data_ptr()
        if self.gemm_split and B1n is not None and (A2 is None or B2n is not None) and K1 % 4 == 0 and (A2 is None or A2.size(1) % 4 == 0):
            _chk2d(B1n, 'B1n')
            assert B1n.shape == (No, K1)
            n2, ld2 = None, 0
            if A2 is not None:
                _chk2d(B2n, 'B2n')
                assert B2n.shape == (No, A2.size(1))
                n2, ld2 = B2n.data_ptr(), B2n.size(1)
            # large products whose B is not registered (qagnn_gemm_nn_prepack_f32): B is split ONCE into the kernel's LDS image order
            # (scratch from the caching allocator, stream-ordered: the next product may reuse it), every row tile then streams it by DMA
            # instead of repeating the split.  The library says whether this very call would use the scratch (0: pre-packed / not taken)
            ws_bytes = self.lib.qagnn_gemm_nn_ws_bytes(C.byref(a), B1n.data_ptr(), K1, n2, ld2) if M >= self.PACK_MIN_M else 0
            if ws_bytes > 0:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A1.device)
                self._check(self.lib.qagnn_gemm_nn_split_ws_f32(C.byref(a), B1n.data_ptr(), K1, n2, ld2, ws.data_ptr(), ws_bytes, self._stream()),
                            'qagnn_gemm_nn_split_ws_f32')
            else:
                self._check(self.lib.qagnn_gemm_nn_split_f32(C.byref(a), B1n.data_ptr(), K1, n2, ld2, self._stream()), 'qagnn_gemm_nn_split_f32')
            return (out, part) if colstats else out
        assert K1 % 16 == 0 and (A2 is None or A2.size(1) % 16 == 0), 'the fp32-MFMA kernel needs K to be a multiple of 16'
        self._check(self.lib.qagnn_gemm_nn_f32(C.byref(a), self._stream()), 'qagnn_gemm_nn_f32')
        return out

    def gemm_tn(self, A, B, a_scale=None, a_shift=None, out=None, accumulate=False, a_rowidx=None, colsum_groups=0,
                b_rowidx=None):
        """C = A^T B.  colsum_groups = G > 0 additionally returns bsum [G, No] = per-group column sums of B."""
        _chk2d(A, 'A'), _chk2d(B, 'B')
        Ka = A.size(1)
        R, No = B.shape
        assert (A.size(0) == R) if a_rowidx is None else (a_rowidx.numel() == R and a_rowidx.dtype == torch.long)
        if out is None:
            assert not accumulate
            out = torch.empty((Ka, No), dtype=torch.float32, device=A.device)
        ws = torch.empty(self.lib.qagnn_gemm_tn_workspace_elems(R, Ka, No), dtype=torch.float32, device=A.device)
        bsum = None
        if colsum_groups:
            assert b_rowidx is None or (b_rowidx.dtype == torch.long and b_rowidx.numel() == R and b_rowidx.is_contiguous())
            bsum = torch.empty((colsum_groups, No), dtype=torch.float32, device=A.device)
        rc = self.lib.qagnn_gemm_tn_colsum_f32(A.data_ptr(), Ka, B.data_ptr(), No, out.data_ptr(), No, R, Ka, No, _ptr(a_scale),
                                               _ptr(a_shift), _ptr(a_rowidx), 1 if accumulate else 0, _ptr(bsum), _ptr(b_rowidx),
                                               colsum_groups, ws.data_ptr(), self._stream())
        self._check(rc, 'qagnn_gemm_tn_colsum_f32')
        return (out, bsum) if colsum_groups else out

    def absmax(self, x, out=None):
        """int32 [1]: the bit pattern of max |x| (qagnn_absmax_f32 into a zeroed word) -- the operand maximum of the three-MFMA GEMM form"""
        assert x.is_contiguous() and x.dtype == torch.float32 and x.numel() % 4 == 0
        if out is None:
            out = torch.empty(4, dtype=torch.int32, device=x.device)
            self._check(self.lib.qagnn_zero_words(out.data_ptr(), 4, self._stream()), 'qagnn_zero_words')
        self._check(self.lib.qagnn_absmax_f32(x.data_ptr(), x.numel(), out.data_ptr(), self._stream()), 'qagnn_absmax_f32')
        return out

    def gemm_tn_h2(self, A1, B, amax_a1, amax_b, A2=None, amax_a2=None, a_scale=None, a_shift=None, out=None):
        """[A1 | A2]^T B in the three-MFMA form (qagnn_gemm_tn_h2_f32); amax_*: absmax() words (amax_a1 AFTER the scale / shift prologue)"""
        _chk2d(A1, 'A1'), _chk2d(B, 'B')
        Ka1, Ka2 = A1.size(1), (A2.size(1) if A2 is not None else 0)
        R, No = B.shape
        if out is None:
            out = torch.empty((Ka1 + Ka2, No), dtype=torch.float32, device=B.device)
        ws = torch.empty(self.lib.qagnn_gemm_tn_workspace_elems(R, Ka1 + Ka2, No), dtype=torch.float32, device=B.device)
        fn = self.lib.qagnn_gemm_tn_h1_f32 if self.gemm_split == 3 else self.lib.qagnn_gemm_tn_h2_f32
        rc = fn(A1.data_ptr(), Ka1, Ka1, _ptr(A2), Ka2, Ka2, B.data_ptr(), No, out.data_ptr(), No, R, No, _ptr(a_scale),
                                           _ptr(a_shift), amax_a1.data_ptr(), _ptr(amax_a2), amax_b.data_ptr(), ws.data_ptr(), self._stream())
        self._check(rc, 'qagnn_gemm_tn_h2_f32')
        return out

    def gemm_tn2(self, A1, A2, B, out=None):
        """[A1 | A2]^T B -> [Ka1 + Ka2, No]: two weight gradients that share their B operand, one launch (qagnn_gemm_tn2_f32)."""
        _chk2d(A1, 'A1'), _chk2d(A2, 'A2'), _chk2d(B, 'B')
        Ka1, Ka2 = A1.size(1), A2.size(1)
        R, No = B.shape
        assert A1.size(0) == R and A2.size(0) == R
        if out is None:
            out = torch.empty((Ka1 + Ka2, No), dtype=torch.float32, device=B.device)
        else:
            _chk2d(out, 'out')
            assert out.shape == (Ka1 + Ka2, No)
        ws = torch.empty(self.lib.qagnn_gemm_tn_workspace_elems(R, Ka1 + Ka2, No), dtype=torch.float32, device=B.device)
        rc = self.lib.qagnn_gemm_tn2_f32(A1.data_ptr(), Ka1, Ka1, A2.data_ptr(), Ka2, Ka2, B.data_ptr(), No, out.data_ptr(), No, R, No,
                                         ws.data_ptr(), self._stream())
        self._check(rc, 'qagnn_gemm_tn2_f32')
        return out

    # -- reductions / elementwise ----------------------------------------------------------------------------------
    def _colreduce(self, mode, X, X2, rowidx, groups, mean, invstd, scale, shift, nout, out_scale=1.0, roww=None, out=None):
        _chk2d(X, 'X')
        R, Cc = X.shape
        if out is None:
            out = torch.empty((nout, Cc), dtype=torch.float32, device=X.device)
        else:
            assert out.shape == (nout, Cc) and out.is_contiguous() and out.dtype == torch.float32
        ws = torch.empty(self.lib.qagnn_colreduce_workspace_elems(R, Cc, groups), dtype=torch.float32, device=X.device)
        rc = self.lib.qagnn_colreduce_f32(mode, X.data_ptr(), Cc, _ptr(X2), Cc, R, Cc, _ptr(rowidx), groups, _ptr(mean),
                                          _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(roww), float(out_scale), out.data_ptr(), ws.data_ptr(), self._stream())
        self._check(rc, 'qagnn_colreduce_f32')
        return out

    def colsum(self, X, rowidx=None, groups=1, scale=1.0, roww=None, out=None):
        return self._colreduce(0, X, None, rowidx, groups, None, None, None, None, groups, scale, roww, out=out)

    def colvar_sum(self, X, mean, scale=1.0, roww=None):
        return self._colreduce(1, X, None, None, 1, mean, None, None, None, 1, scale, roww)[0]

    def bn_finalize(self, mean, var, gamma, beta, eps, running=None, ones_col=-1):
        """-> invstd, scale, shift [Cc]; running = (run_mean [d], run_var [d], num_batches_tracked, dense_pos [d], momentum, unbias)
        additionally applies the train-mode running-statistics update in the same launch.  ones_col >= 0: that (padding) column gets
        scale 0 / shift 1, i.e. relu(bn(h)) carries a column of ones there (see qagnn_bn_finalize_f32)."""
        Cc = mean.numel()
        out = torch.empty((3, Cc), dtype=torch.float32, device=mean.device)
        rm = rv = nbt = pos = None
        d, mom, unb = 0, 0.0, 1.0
        if running is not None:
            rm, rv, nbt, pos, mom, unb = running
            d = rm.numel()
            assert rm.is_contiguous() and rv.is_contiguous() and pos.dtype == torch.long and (nbt is None or nbt.dtype == torch.long)
        rc = self.lib.qagnn_bn_finalize_f32(mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), Cc, _ptr(rm), _ptr(rv), _ptr(nbt),
                                            _ptr(pos), d, float(mom), float(unb), int(ones_col), self._stream())
        self._check(rc, 'qagnn_bn_finalize_f32')
        return out[0], out[1], out[2]

    def bn_stats_finalize(self, part, rows, gamma, beta, eps, running=None, ones_col=-1):
        """part: what gemm_nn(colstats=True) returned for the [rows, Cc] BatchNorm input -> stats [5, Cc] = mean | biased var | invstd |
        scale | shift, plus (running given) the train-mode running-statistics update: qagnn_bn_stats_finalize_f32, one launch."""
        nt, _, Cc = part.shape
        assert part.is_contiguous() and nt == -(-rows // self.STAT_TILE)
        stats = torch.empty((5, Cc), dtype=torch.float32, device=part.device)
        rm = rv = nbt = pos = None
        d, mom, unb = 0, 0.0, 1.0
        if running is not None:
            rm, rv, nbt, pos, mom, unb = running
            d = rm.numel()
            assert rm.is_contiguous() and rv.is_contiguous() and (nbt is None or nbt.dtype == torch.long)
        rc = self.lib.qagnn_bn_stats_finalize_f32(part.data_ptr(), nt, int(rows), Cc, gamma.data_ptr(), beta.data_ptr(), float(eps), stats.data_ptr(),
                                                  _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(pos), d, float(mom), float(unb), int(ones_col), self._stream())
        self._check(rc, 'qagnn_bn_stats_finalize_f32')
        return stats

    def bn_bwd_reduce(self, dR, H, mean, invstd, scale, shift):
        _chk2d(H, 'H')
        return self._colreduce(2, dR, H, None, 1, mean, invstd, scale, shift, 2)

    def bn_relu_bwd(self, dR, H, mean, invstd, scale, shift, gamma, red, inv_rows, roww=None):
        """red = bn_bwd_reduce(...) [2, Cc]; inv_rows = 1/R (batch statistics) or 0 (running statistics); roww [R]: the
        per-row statistics weights when they were not uniform."""
        assert red.is_contiguous() and red.shape == (2, H.size(1))
        _chk2d(dR, 'dR'), _chk2d(H, 'H')
        R, Cc = H.shape
        dH = torch.empty_like(H)
        rc = self.lib.qagnn_bn_relu_bwd_f32(dR.data_ptr(), H.data_ptr(), dH.data_ptr(), Cc, R, Cc, mean.data_ptr(),
                                            invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), gamma.data_ptr(),
                                            red[0].data_ptr(), red[1].data_ptr(), float(inv_rows), _ptr(roww), self._stream())
        self._check(rc, 'qagnn_bn_relu_bwd_f32')
        return dH

    POOL_LIMITS = (4, 256, 1024)  # heads, row width, nodes per subgraph

    def pool_attn_fwd(self, u, cvec, K, mask, inv_temp, p, seed):
        """u [B, NH, Cc], cvec [B, NH], K [B, n, Cc] (contiguous rows), mask [B, n] bool -> attn, attn_d [B, NH, n], z [B, NH, Cc]."""
        B, NH, Cc = u.shape
        n = K.size(1)
        assert u.is_contiguous() and cvec.is_contiguous() and K.is_contiguous() and mask.is_contiguous() and mask.dtype == torch.bool
        attn = torch.empty((2, B, NH, n), dtype=torch.float32, device=u.device)
        z = torch.empty((B, NH, Cc), dtype=torch.float32, device=u.device)
        rc = self.lib.qagnn_pool_attn_fwd_f32(u.data_ptr(), cvec.data_ptr(), K.data_ptr(), K.size(2), mask.data_ptr(), B, n, NH, Cc,
                                              float(inv_temp), float(p), int(seed), attn[0].data_ptr(), attn[1].data_ptr(),
                                              z.data_ptr(), self._stream())
        self._check(rc, 'qagnn_pool_attn_fwd_f32')
        return attn[0], attn[1], z

    def pool_attn_bwd(self, u, K, inv_temp, p, seed, attn, attn_d, dz, dattn_d):
        B, NH, Cc = u.shape
        n = K.size(1)
        assert dz.is_contiguous() and attn.is_contiguous() and attn_d.is_contiguous() and (dattn_d is None or dattn_d.is_contiguous())
        dK = torch.empty_like(K)
        du = torch.empty_like(u)
        dc = torch.empty((B, NH), dtype=torch.float32, device=u.device)
        rc = self.lib.qagnn_pool_attn_bwd_f32(u.data_ptr(), K.data_ptr(), K.size(2), B, n, NH, Cc, float(inv_temp), float(p), int(seed),
                                              attn.data_ptr(), attn_d.data_ptr(), dz.data_ptr(), _ptr(dattn_d), dK.data_ptr(), K.size(2), du.data_ptr(),
                                              dc.data_ptr(), self._stream())
        self._check(rc, 'qagnn_pool_attn_bwd_f32')
        return dK, du, dc

    HEAD_LIMITS = (4, 256, 256)  # heads, NH * dv, DP of qagnn_head_post_{fwd,bwd}_f32

    def head_post_fwd(self, z, attn, BDv, bv, sent, K3, d, w_fc, b_fc, p_pool, p_fc, seed_pool, seed_fc):
        """z [B, NH, DP], attn [B, NH, n] (after dropout), BDv [NH*DP, NH*dv], bv [NH*dv], sent [B, Ds], K3 [B, n, DP] (row 0 of every
        subgraph is read), w_fc [NH*dv + Ds + d], b_fc [1] -> logits [B], out [B, NH*dv] (before dropout), asum [B, NH]."""
        B, NH, DP = z.shape
        n, NO, Ds = attn.size(2), BDv.size(1), sent.size(1)
        for t in (z, attn, BDv, bv, sent, K3, w_fc, b_fc):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert K3.shape == (B, n, DP) and BDv.size(0) == NH * DP and NO % NH == 0 and w_fc.numel() == NO + Ds + d and sent.size(0) == B
        out = torch.empty((B, NO), dtype=torch.float32, device=z.device)
        asum = torch.empty((B, NH), dtype=torch.float32, device=z.device)
        logits = torch.empty((B,), dtype=torch.float32, device=z.device)
        rc = self.lib.qagnn_head_post_fwd_f32(z.data_ptr(), attn.data_ptr(), BDv.data_ptr(), bv.data_ptr(), sent.data_ptr(), K3.data_ptr(), n * DP,
                                              w_fc.data_ptr(), b_fc.data_ptr(), B, NH, DP, NO // NH, n, Ds, d, float(p_pool), float(p_fc),
                                              int(seed_pool), int(seed_fc), out.data_ptr(), asum.data_ptr(), logits.data_ptr(), self._stream())
        self._check(rc, 'qagnn_head_post_fwd_f32')
        return logits, out, asum

    def head_post_bwd(self, dlogits, out, asum, BDv, bv, sent, K3, d, w_fc, p_pool, p_fc, seed_pool, seed_fc, n, need_dsent):
        """-> dz [B, NH, DP], dattn [B, NH, n], dout [B, NH*dv], dsent [B, Ds] or None, dZ [B, DP], part [B, L + NH*dv + 1]"""
        B, NO = out.shape
        NH, DP, Ds = asum.size(1), K3.size(2), sent.size(1)
        assert dlogits.is_contiguous() and dlogits.numel() == B
        dev = out.device
        dz = torch.empty((B, NH, DP), dtype=torch.float32, device=dev)
        dattn = torch.empty((B, NH, n), dtype=torch.float32, device=dev)
        dout = torch.empty((B, NO), dtype=torch.float32, device=dev)
        dsent = torch.empty((B, Ds), dtype=torch.float32, device=dev) if need_dsent else None
        dZ = torch.empty((B, DP), dtype=torch.float32, device=dev)
        part = torch.empty((B, (NO + Ds + d + NO + 1 + 3) // 4 * 4), dtype=torch.float32, device=dev)  # (pitch: a multiple of 4 for the column sums)
        rc = self.lib.qagnn_head_post_bwd_f32(dlogits.data_ptr(), out.data_ptr(), asum.data_ptr(), BDv.data_ptr(), bv.data_ptr(), sent.data_ptr(),
                                              K3.data_ptr(), n * DP, w_fc.data_ptr(), B, NH, DP, NO // NH, n, Ds, d, float(p_pool), float(p_fc),
                                              int(seed_pool), int(seed_fc), dz.data_ptr(), dattn.data_ptr(), dout.data_ptr(), _ptr(dsent),
                                              dZ.data_ptr(), part.data_ptr(), part.size(1), self._stream())
        self._check(rc, 'qagnn_head_post_bwd_f32')
        return dz, dattn, dout, dsent, dZ, part

    GATHER_MAX = GATHER_MAX

    @staticmethod
    def _gather_tabs(tensors):
        t = qagnn_gather_tabs()
        t.n = len(tensors)
        for i, x in enumerate(tensors):
            if x is not None:
                assert x.is_contiguous() and x.dtype == torch.float32
                t.p[i] = x.data_ptr()
        return t

    def gather_multi(self, sources, tid, off):
        """out[i] = sources[tid[i]].flat[off[i]] (tid < 0: 0); tid / off int32 on the device"""
        assert tid.dtype == torch.int32 and off.dtype == torch.int32 and tid.is_contiguous() and off.is_contiguous() and len(sources) <= GATHER_MAX
        out = torch.empty(tid.numel(), dtype=torch.float32, device=tid.device)
        t = self._gather_tabs(sources)
        self._check(self.lib.qagnn_gather_multi_f32(C.byref(t), tid.data_ptr(), off.data_ptr(), out.data_ptr(), tid.numel(), self._stream()),
                    'qagnn_gather_multi_f32')
        return out

    def gather_multi_sum(self, grads, tid, off):
        """out[s] = sum_k grads[tid[k, s]].flat[off[k, s]] (tid < 0 or an absent gradient: 0); tid / off [K, S] int32"""
        assert tid.dtype == torch.int32 and off.dtype == torch.int32 and tid.is_contiguous() and off.is_contiguous() and tid.dim() == 2
        assert len(grads) <= GATHER_MAX
        out = torch.empty(tid.size(1), dtype=torch.float32, device=tid.device)
        t = self._gather_tabs(grads)
        self._check(self.lib.qagnn_gather_multi_sum_f32(C.byref(t), tid.data_ptr(), off.data_ptr(), tid.size(0), tid.size(1), out.data_ptr(),
                                                        self._stream()), 'qagnn_gather_multi_sum_f32')
        return out

    def add_row0(self, dK, dZ):
        """dK [B, n, Cc] (contiguous): dK[:, 0, :] += dZ [B, Cc], in place."""
        B, n, Cc = dK.shape
        assert dK.is_contiguous() and dZ.is_contiguous() and dZ.shape == (B, Cc)
        self._check(self.lib.qagnn_add_row0_f32(dK.data_ptr(), n * Cc, dZ.data_ptr(), B, Cc, self._stream()), 'qagnn_add_row0_f32')
        return dK

    def timing_enable(self, on):
        """Library-side HIP-event brackets around the GEMM and edge-stage entry points (qagnn_timing_enable): measurement harnesses only."""
        self._check(self.lib.qagnn_timing_enable(1 if on else 0), 'qagnn_timing_enable')

    def timing_read(self):
        """-> {kind: (milliseconds, calls)} for 'gemm_nn', 'gemm_tn', 'edge_attn_fwd', 'edge_attn_bwd' since timing_enable(True)"""
        ms, calls = (C.c_double * 4)(), (C.c_int64 * 4)()
        self._check(self.lib.qagnn_timing_read(ms, calls), 'qagnn_timing_read')
        return {k: (ms[i], calls[i]) for i, k in enumerate(('gemm_nn', 'gemm_tn', 'edge_attn_fwd', 'edge_attn_bwd'))}

    def gelu_dropout_fwd(self, X, p, seed, amax=False):
        """amax=True: -> (Y, word) with word = int32 [4], [0] = the bit pattern of max |Y| (qagnn_gelu_dropout_fwd_amax_f32)"""
        assert X.is_contiguous() and X.dtype == torch.float32
        Y = torch.empty_like(X)
        if amax:
            word = torch.empty(4, dtype=torch.int32, device=X.device)
            self._check(self.lib.qagnn_zero_words(word.data_ptr(), 4, self._stream()), 'qagnn_zero_words')
            scratch = torch.empty(self.lib.qagnn_gelu_dropout_amax_scratch_elems(X.numel()), dtype=torch.float32, device=X.device)
            self._check(self.lib.qagnn_gelu_dropout_fwd_amax_f32(X.data_ptr(), Y.data_ptr(), X.numel(), float(p), int(seed), word.data_ptr(),
                                                                 scratch.data_ptr(), self._stream()), 'qagnn_gelu_dropout_fwd_amax_f32')
            return Y, word
        self._check(self.lib.qagnn_gelu_dropout_fwd_f32(X.data_ptr(), Y.data_ptr(), X.numel(), float(p), int(seed),
                                                        self._stream()), 'qagnn_gelu_dropout_fwd_f32')
        return Y

    def gelu_dropout_bwd(self, X, dY, p, seed, amax=False):
        """amax=True: -> (dX, word) with word = int32 [4], [0] = the bit pattern of max |dX| (qagnn_gelu_dropout_bwd_amax_f32)"""
        assert X.is_contiguous() and dY.is_contiguous()
        dX = torch.empty_like(X)
        if amax:
            word = torch.empty(4, dtype=torch.int32, device=X.device)
            self._check(self.lib.qagnn_zero_words(word.data_ptr(), 4, self._stream()), 'qagnn_zero_words')
            scratch = torch.empty(self.lib.qagnn_gelu_dropout_amax_scratch_elems(X.numel()), dtype=torch.float32, device=X.device)
            self._check(self.lib.qagnn_gelu_dropout_bwd_amax_f32(X.data_ptr(), dY.data_ptr(), dX.data_ptr(), X.numel(), float(p), int(seed),
                                                                 word.data_ptr(), scratch.data_ptr(), self._stream()), 'qagnn_gelu_dropout_bwd_amax_f32')
            return dX, word
        self._check(self.lib.qagnn_gelu_dropout_bwd_f32(X.data_ptr(), dY.data_ptr(), dX.data_ptr(), X.numel(), float(p),
                                                        int(seed), self._stream()), 'qagnn_gelu_dropout_bwd_f32')
        return dX

    def bn_relu_bwd_colsum(self, dR, H, mean, invstd, scale, shift, gamma, red, inv_rows, roww=None):
        """bn_relu_bwd that also returns colsum(dH) [Cc]."""
        assert red.is_contiguous() and red.shape == (2, H.size(1))
        _chk2d(dR, 'dR'), _chk2d(H, 'H')
        R, Cc = H.shape
        dH = torch.empty_like(H)
        cs = torch.empty(Cc, dtype=torch.float32, device=H.device)
        ws = torch.empty(self.lib.qagnn_colreduce_workspace_elems(R, Cc, 1), dtype=torch.float32, device=H.device)
        rc = self.lib.qagnn_bn_relu_bwd_colsum_f32(dR.data_ptr(), H.data_ptr(), dH.data_ptr(), Cc, R, Cc, mean.data_ptr(), invstd.data_ptr(),
                                                   scale.data_ptr(), shift.data_ptr(), gamma.data_ptr(), red[0].data_ptr(), red[1].data_ptr(),
                                                   float(inv_rows), _ptr(roww), cs.data_ptr(), ws.data_ptr(), self._stream())
        self._check(rc, 'qagnn_bn_relu_bwd_colsum_f32')
        return dH, cs

    def sin_basis(self, score, js, ldo):
        assert score.is_contiguous() and js.is_contiguous() and score.dtype == js.dtype == torch.float32
        R, J = score.numel(), js.numel()
        out = torch.empty((R, ldo), dtype=torch.float32, device=score.device)
        self._check(self.lib.qagnn_sin_basis_f32(score.data_ptr(), js.data_ptr(), out.data_ptr(), ldo, R, J, self._stream()),
                    'qagnn_sin_basis_f32')
        return out

    # -- edge kernels -------------------------------------------------------------------------------------------------
    def edge_attn_fwd(self, graph, KMQ, EkEm, HP, qscale):
        _chk2d(KMQ, 'KMQ'), _chk2d(EkEm, 'EkEm')
        DP = 4 * HP
        assert KMQ.shape == (graph.N, 3 * DP) and EkEm.shape == (graph.C, 2 * DP)
        dev = KMQ.device
        a = torch.empty((graph.Ep, 4), dtype=torch.float32, device=dev)
        alpha = torch.empty_like(a)
        aggr = torch.empty((graph.N, DP), dtype=torch.float32, device=dev)
        score = torch.empty_like(a)  # scratch of the generic kernels (raw scores of hub segments)
        rc = self.lib.qagnn_edge_attn_fwd_f32(C.byref(graph.c), KMQ.data_ptr(), 3 * DP, EkEm.data_ptr(), 2 * DP, HP, float(qscale), score.data_ptr(),
                                              a.data_ptr(), alpha.data_ptr(), aggr.data_ptr(), DP, self._stream())
        self._check(rc, 'qagnn_edge_attn_fwd_f32')
        return aggr, a, alpha

    def edge_attn_bwd(self, graph, KMQ, EkEm, HP, qscale, a, alpha, G):
        _chk2d(KMQ, 'KMQ'), _chk2d(EkEm, 'EkEm'), _chk2d(G, 'G')
        DP = 4 * HP
        dev = KMQ.device
        dKMQ = torch.empty_like(KMQ)
        dEkEm = torch.empty_like(EkEm)
        ga = torch.empty((graph.Ep, 4), dtype=torch.float32, device=dev)
        rs = torch.empty((graph.N, 4), dtype=torch.float32, device=dev)
        cls_part = torch.empty((graph.max_chunks + CLS_SLICES * graph.C, 2 * DP), dtype=torch.float32, device=dev)
        rc = self.lib.qagnn_edge_attn_bwd_f32(C.byref(graph.c), KMQ.data_ptr(), 3 * DP, EkEm.data_ptr(), 2 * DP, HP,
                                              float(qscale), a.data_ptr(), alpha.data_ptr(), G.data_ptr(), DP,
                                              dKMQ.data_ptr(), dEkEm.data_ptr(), ga.data_ptr(), rs.data_ptr(),
                                              cls_part.data_ptr(), self._stream())
        self._check(rc, 'qagnn_edge_attn_bwd_f32')
        return dKMQ, dEkEm

    # -- one GATConvE hop per call (csrc/hop.hip) ---------------------------------------------------------------------------
    def _hop_struct(self, graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, tab_col=-1, side=False):
        Wx_t, Wx, Ws_t, Ws, TT, EkEm, W1t, W1, b1, gamma, beta, W2t, W2, b2, run_mean_p, run_var_p = prm
        DP = 4 * HP
        _chk2d(X, 'X'), _chk2d(Wx_t, 'Wx_t'), _chk2d(Wx, 'Wx'), _chk2d(TT, 'TT'), _chk2d(EkEm, 'EkEm')
        for t in (W1t, W1, W2t, W2):
            _chk2d(t, 'mlp weight')
        for t in (b1, gamma, beta, b2, run_mean_p, run_var_p):
            assert t.is_contiguous() and t.numel() == DP and t.dtype == torch.float32
        assert X.shape == (graph.N, DP) and Wx_t.shape == (DP, 3 * DP) and Wx.shape == (3 * DP, DP) and EkEm.shape == (graph.C, 2 * DP)
        assert ntype.dtype == torch.long and ntype.numel() == graph.N and ntype.is_contiguous() and TT.size(1) == 3 * DP
        h = qagnn_hop_args()
        h.g = C.pointer(graph.c)
        h.N, h.DP, h.HP, h.T, h.qscale = graph.N, DP, HP, TT.size(0), float(qscale)
        h.X, h.ntype = X.data_ptr(), ntype.data_ptr()
        if S is not None:
            _chk2d(S, 'S'), _chk2d(Ws_t, 'Ws_t'), _chk2d(Ws, 'Ws')
            SP = S.size(1)
            assert S.size(0) == graph.N and Ws_t.shape == (SP, 3 * DP) and Ws.shape == (3 * DP, SP)
            h.SP, h.S, h.Ws_t, h.Ws = SP, S.data_ptr(), Ws_t.data_ptr(), Ws.data_ptr()
        h.Wx_t, h.Wx, h.TT, h.EkEm = Wx_t.data_ptr(), Wx.data_ptr(), TT.data_ptr(), EkEm.data_ptr()
        h.W1t, h.W1, h.b1, h.gamma, h.beta = W1t.data_ptr(), W1.data_ptr(), b1.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        h.W2t, h.W2, h.b2 = W2t.data_ptr(), W2.data_ptr(), b2.data_ptr()
        h.batch_stats, h.eps = (1 if batch_stats else 0), float(eps)
        h.run_mean_p, h.run_var_p = run_mean_p.data_ptr(), run_var_p.data_ptr()
        h.apply_act, h.p_drop, h.seed = (1 if apply_act else 0), float(p), int(seed)
        h.gemm_split = int(self.gemm_split)
        tc, oc = tab_col if isinstance(tab_col, tuple) else (tab_col, -1)  # (type-indicator column of S, ones column of relu(bn(h1)))
        h.tab_col = int(tc) if S is not None else -1
        h.ones_col = int(oc)
        h.side_stream = self._side_stream() if side else None  # backward only (qagnn_hop_args.side_stream)
        return h

    def hop_fwd(self, graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, running, cols=-1):
        """-> (y, saved) with saved = (KMQ, aa [2, Ep, 4] = a | alpha, aggr, h1, out, stats [5, DP]); y is `out` when not apply_act."""
        h = self._hop_struct(graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, cols)
        N, DP, dev = graph.N, 4 * HP, X.device
        KMQ = torch.empty((N, 3 * DP), dtype=torch.float32, device=dev)
        aa = torch.empty((2, graph.Ep, 4), dtype=torch.float32, device=dev)
        rows = torch.empty((4 if apply_act else 3, N, DP), dtype=torch.float32, device=dev)  # aggr, h1, out (, y)
        stats = torch.empty((5, DP), dtype=torch.float32, device=dev)
        amax = torch.empty(HOP_AMAX_WORDS, dtype=torch.int32, device=dev)  # operand maxima of the three-MFMA form (zeroed by the library)
        h.amax = amax.data_ptr()
        ws = torch.empty(self.lib.qagnn_hop_fwd_workspace_elems(N, graph.Ep, DP), dtype=torch.float32, device=dev)
        h.KMQ, h.a, h.alpha, h.stats = KMQ.data_ptr(), aa[0].data_ptr(), aa[1].data_ptr(), stats.data_ptr()
        h.aggr, h.h1, h.out = rows[0].data_ptr(), rows[1].data_ptr(), rows[2].data_ptr()
        if apply_act:
            h.y = rows[3].data_ptr()
        if running is not None:
            rm, rv, nbt, pos, mom, _unb = running
            assert rm.is_contiguous() and rv.is_contiguous() and pos.dtype == torch.long and (nbt is None or nbt.dtype == torch.long)
            h.run_mean, h.run_var, h.num_batches_tracked, h.dense_pos = rm.data_ptr(), rv.data_ptr(), _ptr(nbt), pos.data_ptr()
            h.d, h.momentum = rm.numel(), float(mom)
        h.ws, h.ws_elems = ws.data_ptr(), ws.numel()
        self._check(self.lib.qagnn_hop_fwd_f32(C.byref(h), self._stream()), 'qagnn_hop_fwd_f32')
        return rows[3 if apply_act else 2], (KMQ, aa, rows[0], rows[1], rows[2], stats, amax)

    def hop_bwd(self, graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, saved, dy, need_dX, need_dS,
                dX_acc=None, dS_acc=None, tab_col=-1, overlap=True):
        """-> (dX, dS, dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dgamma, dbeta, dW2t, db2); overlap: the weight-gradient products on a side stream"""
        h = self._hop_struct(graph, HP, qscale, X, S, ntype, prm, batch_stats, eps, p, seed, apply_act, tab_col, side=overlap)
        KMQ, aa, aggr, h1, out, stats = saved[:6]
        h.amax = saved[6].data_ptr() if len(saved) > 6 else None
        N, DP, dev, SP, T = graph.N, 4 * HP, X.device, h.SP, h.T
        _chk2d(dy, 'dy')
        h.KMQ, h.a, h.alpha, h.stats = KMQ.data_ptr(), aa[0].data_ptr(), aa[1].data_ptr(), stats.data_ptr()
        h.aggr, h.h1, h.out, h.y = aggr.data_ptr(), h1.data_ptr(), out.data_ptr(), out.data_ptr()
        h.dy = dy.data_ptr()
        sizes = [DP * 3 * DP, SP * 3 * DP, T * 3 * DP, graph.C * 2 * DP, DP * DP, DP, 2 * DP, DP * DP, DP]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)  # every size is a multiple of 4: 16-byte aligned views
        parts, off = [], 0
        for n in sizes:
            parts.append(flat[off:off + n])
            off += n
        dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dbn, dW2t, db2 = parts
        h.dWx_t, h.dTT, h.dEkEm, h.dW1t, h.db1 = dWx_t.data_ptr(), dTT.data_ptr(), dEkEm.data_ptr(), dW1t.data_ptr(), db1.data_ptr()
        h.dbn, h.dW2t, h.db2 = dbn.data_ptr(), dW2t.data_ptr(), db2.data_ptr()
        if h.ones_col >= 0:  # the bias gradient IS that row of dW2t: hand out the view, no copy
            db2 = dW2t.view(DP, DP)[h.ones_col]
            h.db2 = db2.data_ptr()
        if SP and h.tab_col >= 0:  # likewise the type-table gradient: rows of dWs_t
            dTT = dWs_t.view(SP, 3 * DP)[h.tab_col:h.tab_col + T]
            h.dTT = dTT.data_ptr()
        dX = dS = None
        if need_dX:  # *_acc: an existing running total of this gradient, added to in place (GEMM epilogue accumulate)
            dX = dX_acc if dX_acc is not None else torch.empty((N, DP), dtype=torch.float32, device=dev)
            assert dX.shape == (N, DP) and dX.is_contiguous()
            h.dX, h.accumulate_dX = dX.data_ptr(), (1 if dX_acc is not None else 0)
        if SP:
            h.dWs_t = dWs_t.data_ptr()
            if need_dS:
                dS = dS_acc if dS_acc is not None else torch.empty((N, SP), dtype=torch.float32, device=dev)
                assert dS.shape == (N, SP) and dS.is_contiguous()
                h.dS, h.accumulate_dS = dS.data_ptr(), (1 if dS_acc is not None else 0)
        ws = torch.empty(self.lib.qagnn_hop_bwd_workspace_elems(N, graph.Ep, DP, SP, graph.max_chunks + CLS_SLICES * graph.C), dtype=torch.float32, device=dev)
        h.ws, h.ws_elems = ws.data_ptr(), ws.numel()
        self._check(self.lib.qagnn_hop_bwd_f32(C.byref(h), self._stream()), 'qagnn_hop_bwd_f32')
        return (dX, dS, dWx_t.view(DP, 3 * DP), dWs_t.view(SP, 3 * DP) if SP else None, dTT.view(T, 3 * DP), dEkEm.view(graph.C, 2 * DP),
                dW1t.view(DP, DP), db1, dbn[DP:], dbn[:DP], dW2t.view(DP, DP), db2)

    # -- the whole k-hop stack per call (csrc/hop.hip: qagnn_stack_{fwd,bwd}_f32) -------------------------------------------------------
    def stack_fwd(self, graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, runnings, cols=-1, x_amax=None, s_amax=None):
        """k hops with GELU + dropout after each; prms / seeds / runnings: per-layer lists.  -> (y [N, DP], saved)."""
        k = len(prms)
        N, DP, dev = graph.N, 4 * HP, X.device
        KMQ = torch.empty((k, N, 3 * DP), dtype=torch.float32, device=dev)
        aa = torch.empty((k, 2, graph.Ep, 4), dtype=torch.float32, device=dev)
        rows = torch.empty((k, 4, N, DP), dtype=torch.float32, device=dev)  # per hop: aggr, h1, out, y
        stats = torch.empty((k, 5, DP), dtype=torch.float32, device=dev)
        amax = torch.empty((k, HOP_AMAX_WORDS), dtype=torch.int32, device=dev)  # one array: the library zeroes it with one launch
        p_amax = amax.data_ptr()
        ws = torch.empty(self.lib.qagnn_hop_fwd_workspace_elems(N, graph.Ep, DP), dtype=torch.float32, device=dev)
        hops = (qagnn_hop_args * k)()
        x = X
        # addresses by arithmetic: a view tensor per pointer costs the host-bound batches ~0.3 ms per step
        p_kmq, p_aa, p_rows, p_stats, row_b = KMQ.data_ptr(), aa.data_ptr(), rows.data_ptr(), stats.data_ptr(), N * DP * 4
        for l in range(k):
            h = self._hop_struct(graph, HP, qscale, x, S, ntype, prms[l], batch_stats, eps, p, seeds[l], True, cols)
            h.KMQ, h.stats = p_kmq + l * 3 * row_b, p_stats + l * 5 * DP * 4
            h.a, h.alpha = p_aa + (2 * l) * graph.Ep * 16, p_aa + (2 * l + 1) * graph.Ep * 16
            h.aggr, h.h1, h.out, h.y = (p_rows + (4 * l + i) * row_b for i in range(4))
            h.amax = p_amax + l * HOP_AMAX_WORDS * 4
            if l == 0 and x_amax is not None:  # max |X| from X's producer (ops.amax_lookup): no reduction pass over the stack input
                h.x_amax = x_amax.data_ptr()
            if s_amax is not None:
                h.s_amax = s_amax.data_ptr()
            if runnings[l] is not None:
                rm, rv, nbt, pos, mom, _unb = runnings[l]
                assert rm.is_contiguous() and rv.is_contiguous() and pos.dtype == torch.long and (nbt is None or nbt.dtype == torch.long)
                h.run_mean, h.run_var, h.num_batches_tracked, h.dense_pos = rm.data_ptr(), rv.data_ptr(), _ptr(nbt), pos.data_ptr()
                h.d, h.momentum = rm.numel(), float(mom)
            h.ws, h.ws_elems = ws.data_ptr(), ws.numel()
            hops[l] = h
            x = rows[l, 3]
        self._check(self.lib.qagnn_stack_fwd_f32(hops, k, self._stream()), 'qagnn_stack_fwd_f32')
        ext = torch.empty(0, dtype=torch.int32, device=dev)
        return rows[k - 1, 3], (KMQ, aa, rows, stats, amax, x_amax if x_amax is not None else ext, s_amax if s_amax is not None else ext)

    def stack_bwd(self, graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, saved, dy, need_dX, need_dS, dX_acc=None, tab_col=-1,
                  overlap=True):
        """-> (dX, dS, [per layer: (dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dgamma, dbeta, dW2t, db2)]); dX_acc: an existing running total of
        the stack input's gradient (the output GEMM's share), added to in place."""
        k = len(prms)
        KMQ, aa, rows, stats = saved[:4]
        p_amax = saved[4].data_ptr() if len(saved) > 4 else None
        x_amax, s_amax = (saved[5], saved[6]) if len(saved) > 6 else (None, None)
        N, DP, dev = graph.N, 4 * HP, X.device
        SP = S.size(1) if S is not None else 0
        T = prms[0][4].size(0)
        _chk2d(dy, 'dy')
        sizes = [DP * 3 * DP, SP * 3 * DP, T * 3 * DP, graph.C * 2 * DP, DP * DP, DP, 2 * DP, DP * DP, DP]
        per = sum(sizes)
        flat = torch.empty(k * per, dtype=torch.float32, device=dev)  # every size is a multiple of 4: 16-byte aligned views
        dxs = torch.empty((max(k - 1, 1), N, DP), dtype=torch.float32, device=dev)  # gradient handed from hop l to hop l-1
        dS = torch.empty((N, SP), dtype=torch.float32, device=dev) if (SP and need_dS) else None
        dX = None
        if need_dX:
            dX = dX_acc if dX_acc is not None else torch.empty((N, DP), dtype=torch.float32, device=dev)
            assert dX.shape == (N, DP) and dX.is_contiguous()
        ws = torch.empty(self.lib.qagnn_hop_bwd_workspace_elems(N, graph.Ep, DP, SP, graph.max_chunks + CLS_SLICES * graph.C), dtype=torch.float32, device=dev)
        hops = (qagnn_hop_args * k)()
        grads = []
        p_kmq, p_aa, p_rows, p_stats, row_b = KMQ.data_ptr(), aa.data_ptr(), rows.data_ptr(), stats.data_ptr(), N * DP * 4
        p_flat, p_dxs = flat.data_ptr(), dxs.data_ptr()
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        for l in range(k):
            x = X if l == 0 else rows[l - 1, 3]
            h = self._hop_struct(graph, HP, qscale, x, S, ntype, prms[l], batch_stats, eps, p, seeds[l], True, tab_col, side=overlap)
            h.KMQ, h.stats = p_kmq + l * 3 * row_b, p_stats + l * 5 * DP * 4
            h.a, h.alpha = p_aa + (2 * l) * graph.Ep * 16, p_aa + (2 * l + 1) * graph.Ep * 16
            h.aggr, h.h1, h.out = (p_rows + (4 * l + i) * row_b for i in range(3))
            h.y = p_rows + (4 * l + 3) * row_b  # (what the forward wrote: the chain hops[l + 1].X == hops[l].y is what shares the amax words)
            h.amax = p_amax + l * HOP_AMAX_WORDS * 4 if p_amax else None
            if l == 0 and x_amax is not None and x_amax.numel():
                h.x_amax = x_amax.data_ptr()
            if s_amax is not None and s_amax.numel():
                h.s_amax = s_amax.data_ptr()
            h.dy = dy.data_ptr() if l == k - 1 else p_dxs + l * row_b
            base = p_flat + l * per * 4
            h.dWx_t, pdWs_t, h.dTT, h.dEkEm, h.dW1t, h.db1, h.dbn, h.dW2t, h.db2 = (base + o * 4 for o in offs[:9])
            fl = flat[l * per:(l + 1) * per]
            dWx_t, dWs_t, dTT, dEkEm, dW1t, db1, dbn, dW2t, db2 = fl.split(sizes)
            if h.ones_col >= 0:  # the bias gradient IS that row of dW2t, the type-table gradient those rows of dWs_t: views, no copies
                db2 = dW2t.view(DP, DP)[h.ones_col]
                h.db2 = h.dW2t + h.ones_col * DP * 4
            if SP and h.tab_col >= 0:
                dTT = dWs_t.view(SP, 3 * DP)[h.tab_col:h.tab_col + T]
                h.dTT = pdWs_t + h.tab_col * 3 * DP * 4
            if l > 0:
                h.dX, h.accumulate_dX = p_dxs + (l - 1) * row_b, 0
            elif dX is not None:
                h.dX, h.accumulate_dX = dX.data_ptr(), (1 if dX_acc is not None else 0)
            if SP:
                h.dWs_t = pdWs_t
                if dS is not None:
                    h.dS, h.accumulate_dS = dS.data_ptr(), (0 if l == k - 1 else 1)  # hop k-1's backward runs first
            h.ws, h.ws_elems = ws.data_ptr(), ws.numel()
            hops[l] = h
            grads.append((dWx_t.view(DP, 3 * DP), dWs_t.view(SP, 3 * DP) if SP else None, dTT.view(T, 3 * DP), dEkEm.view(graph.C, 2 * DP),
                          dW1t.view(DP, DP), db1, dbn[DP:], dbn[:DP], dW2t.view(DP, DP), db2))
        self._check(self.lib.qagnn_stack_bwd_f32(hops, k, self._stream()), 'qagnn_stack_bwd_f32')
        return dX, dS, grads
