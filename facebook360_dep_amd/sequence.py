"""Frames of a sequence sharded over GPUs: host-side mirror of `derp_seq_*` (include/derp_hip.h).

The schedule is the per-level barrier of the reference's render pipeline
(scripts/render/pipeline.py:364-408): for level L, coarse to fine,
    DerpCLI(level L) on every frame  ->  TemporalBilateralFilter(level L) over [t-R, t+R]
    ->  "Transfer": the filtered level overwrites disparity_levels/level_L  ->  level L-1.
The reference moves the +-R frames through the filesystem (TemporalBilateralFilter.cpp:139-160).
Here a rank owns a contiguous chunk of frames (render.py:169-175), and the only data that crosses
ranks inside the level loop is the raw level-L disparity of the frames a neighbour's window reaches
into (the halo). Partition, windows and the transfer plan come from the C library's host-only
functions, so this module, the C++ driver and the tests agree by construction.

`SequenceRunner` drives the HIP library. Transports, tried in this order by `attach_best`:
  "rccl"      ncclSend/ncclRecv issued by the library itself on its own stream (no torch in the loop)
  "torch"     the same plan moved with torch.distributed isend/irecv over device-pointer views
  "broadcast" one torch.distributed broadcast per transferred frame (collective fallback)
`run_schedule` is the transport-agnostic loop; the CPU tests run it with gloo and the oracle as the
compute backend.
"""
import ctypes as C
import weakref

from . import derp

BLOCK, CYCLIC = 0, 1
KIND_COLOR, KIND_FG, KIND_DISPARITY = 0, 1, 2


def disparity_crc(plane, crc=0):
    """zlib CRC-32 of a float32 disparity plane (NaN payloads canonicalised), chained from `crc`."""
    import zlib

    import numpy as np

    a = np.ascontiguousarray(plane, np.float32)
    a = np.where(np.isnan(a), np.float32(np.nan), a).astype(np.float32)
    return zlib.crc32(a.tobytes(), crc) & 0xFFFFFFFF


# ---------------------------------------------------------------- host-only plan (no GPU needed)
def temporal_window(t, first, last, radius):
    """populateMinMaxFrame (TemporalBilateralFilter.cpp:96-119): frames that exist in
    [t - radius, t + radius], i.e. the window clamped to the sequence [first, last]."""
    lo, hi = C.c_int(), C.c_int()
    derp.lib().derp_seq_window(t, first, last, radius, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def owner(first, last, world, frame, partition=BLOCK):
    return derp.lib().derp_seq_owner(first, last, world, partition, frame)


def owned_frames(first, last, world, rank, partition=BLOCK):
    return [t for t in range(first, last + 1) if owner(first, last, world, t, partition) == rank]


def plan(first, last, world, radius, partition=BLOCK):
    """-> [(frame, from_rank, to_rank)]: every transfer of one exchange, in the order all ranks post them."""
    n = derp.lib().derp_seq_plan(first, last, world, radius, partition, None, 0)
    if n < 0:
        raise ValueError("bad sequence geometry")
    buf = (derp.SeqTransfer * max(n, 1))()
    derp.lib().derp_seq_plan(first, last, world, radius, partition, buf, n)
    return [(buf[i].frame, buf[i].from_rank, buf[i].to_rank) for i in range(n)]


def halo_frames(first, last, world, rank, radius, partition=BLOCK):
    return sorted({f for (f, a, b) in plan(first, last, world, radius, partition) if b == rank})


# ---------------------------------------------------------------- transport-agnostic exchange
def exchange(transfers, rank, tensor_of, dist, mode="p2p", scratch=None):
    """Move frames along `transfers`. tensor_of(frame) -> contiguous tensor: the owned frame's buffer on
    the sender, the halo buffer to fill on the receiver. Returns the bytes this rank received."""
    received = 0
    if mode == "p2p":
        ops = []
        for (f, a, b) in transfers:
            if a == rank:
                ops.append(dist.P2POp(dist.isend, tensor_of(f), b))
            elif b == rank:
                t = tensor_of(f)
                received += t.numel() * t.element_size()
                ops.append(dist.P2POp(dist.irecv, t, a))
        if ops:
            for work in dist.batch_isend_irecv(ops):
                work.wait()
        return received
    # collective fallback: one broadcast per transferred frame; ranks outside the frame's halo take
    # part with a scratch buffer of the same size (`scratch()`)
    spare = None
    done = set()
    for (f, a, b) in transfers:
        if f in done:
            continue
        done.add(f)
        if a == rank or any(bb == rank for (ff, aa, bb) in transfers if ff == f):
            t = tensor_of(f)
            if a != rank:
                received += t.numel() * t.element_size()
        else:
            spare = scratch() if spare is None else spare
            t = spare
        dist.broadcast(t, src=a)
    return received


def run_schedule(backend, levels, first, last, rank, world, radius=2, partition=BLOCK, dist=None, mode="p2p"):
    """The per-level barrier for this rank's frames with an EXTERNAL transport (torch.distributed).

    backend.compute(level)                 processLevel of every owned frame
    backend.tensor(frame, level, kind)     buffer of an owned / halo frame (see `exchange`)
    backend.scratch(level, kind)           same-sized spare buffer (collective fallback only)
    backend.before_exchange() / after_exchange()   stream fences around foreign-stream traffic
    backend.filter(level)                  temporal filter of every owned frame + Transfer
    """
    transfers = plan(first, last, world, radius, partition) if world > 1 else []
    received = 0
    for level in levels:
        backend.compute(level)
        if transfers:
            backend.before_exchange()
            received += exchange(transfers, rank, lambda f: backend.tensor(f, level, KIND_DISPARITY), dist, mode,
                                 lambda: backend.scratch(level, KIND_DISPARITY))
            backend.after_exchange()
            getattr(backend, "exchanged", lambda lv: None)(level)
        backend.filter(level)
    return received


# ---------------------------------------------------------------- HIP-backed runner
class SequenceRunner:
    """derp_seq over one `derp.Derp` context: owns the frame slots of this rank's frames."""

    def __init__(self, g, first, last, rank=0, world=1, **options):
        self.g, self.first, self.last, self.rank, self.world = g, first, last, rank, world
        self.opt = derp.SeqOptions()
        derp.lib().derp_seq_options_default(C.byref(self.opt))
        for k, v in options.items():
            if not hasattr(self.opt, k):
                raise KeyError(k)
            setattr(self.opt, k, type(getattr(self.opt, k))(v))
        h = C.c_void_p()
        g._ck(derp.lib().derp_seq_create(C.byref(h), g.h, first, last, rank, world, C.byref(self.opt)))
        self.h = h
        g._seqs.append(weakref.ref(self))
        has_transfers = world > 1 and bool(plan(first, last, world, self.opt.time_radius, self.opt.partition)) \
            and bool(self.opt.do_temporal_filter)
        self.transport = "none" if has_transfers else "local"  # nothing to move: replicas
        n_owned, n_halo = C.c_int(), C.c_int()
        derp.lib().derp_seq_counts(self.h, C.byref(n_owned), C.byref(n_halo))
        self.owned = self._frames(0, n_owned.value)
        self.halo = self._frames(1, n_halo.value)
        self._dist = None
        self._mode = "p2p"
        self._stage = False
        self._pending = []
        self._received = 0  # bytes moved by the external (torch.distributed) transports
        # out of core (resident_frames < owned frames): inputs stay in host memory, kept alive here
        self.streaming = 0 < self.opt.resident_frames < len(self.owned)
        self._host = {}

    def _frames(self, halo, n):
        buf = (C.c_int * max(n, 1))()
        derp.lib().derp_seq_frames(self.h, halo, buf, n)
        return [buf[i] for i in range(n)]

    def close(self):
        if getattr(self, "h", None):
            derp.lib().derp_seq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        self.g._ck(rc)

    # ---- inputs
    def slot(self, frame):
        return derp.lib().derp_seq_frame_slot(self.h, frame)

    def upload_frame(self, frame, data):
        """Upload `data` (synth.make_frame dict) as sequence frame `frame` (must be owned). Out of core the
        library streams from the host arrays level by level; they are kept alive by this object."""
        import numpy as np

        s = self.slot(frame)
        if s < 0:
            raise ValueError("frame %d is not owned by rank %d" % (frame, self.rank))
        if not self.streaming:
            self.g.select_frame(s)
            self.g.upload_frame(data)
            return

        def arr(x):
            return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

        g = self.g
        keep = []
        for level in range(len(g.sizes)):
            col = np.ascontiguousarray(np.stack([arr(data["color"][level][s]) for s in range(g.S)]), np.uint16)
            fg = bg = None
            if data.get("masks"):
                fg = np.ascontiguousarray(np.stack([arr(data["masks"][level][s]) for s in range(g.S)]), np.uint8)
            if data.get("bg_disp"):
                bg = np.ascontiguousarray(np.stack([arr(data["bg_disp"][level][g._dst_src(d)]) for d in range(g.D)]),
                                          np.float32)
            keep.append((col, fg, bg))
            self._ck(derp.lib().derp_seq_host_inputs(
                self.h, frame, level, col.ctypes.data_as(C.c_void_p),
                fg.ctypes.data_as(C.c_void_p) if fg is not None else None,
                bg.ctypes.data_as(C.c_void_p) if bg is not None else None))
        self._host[frame] = keep

    def download_disparity(self, frame, level, d):
        import numpy as np

        w, h = self.g.sizes[level]
        out = np.zeros((h, w), dtype=np.float32)
        self._ck(derp.lib().derp_seq_download_disparity(self.h, frame, level, d, out.ctypes.data_as(C.c_void_p)))
        return out

    def result_crc(self, level=0):
        """{frame: CRC-32 of the frame's level-`level` disparity, destinations in rig order, float32 bytes with
        every NaN written as 0x7fc00000} for every owned frame — what a multi-GPU run prints so that it can be
        checked against the 1-GPU run."""
        out = {}
        for t in self.owned:
            crc = 0
            for d in range(self.g.D):
                crc = disparity_crc(self.download_disparity(t, level, d), crc)
            out[t] = crc
        return out

    def buffer(self, frame, level, kind):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(derp.lib().derp_seq_buffer(self.h, frame, level, kind, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _device_view(self, frame, level, kind):
        import torch

        p, n = self.buffer(frame, level, kind)

        class _A:
            pass

        a = _A()
        a.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 3}
        return torch.as_tensor(a, device=torch.device("cuda", torch.cuda.current_device()))

    def tensor(self, frame, level, kind):
        """uint8 torch view of an owned / halo frame's device buffer — or, when the backend cannot move
        device memory (gloo), a host copy to send / a pinned host buffer to receive into, copied to the
        device buffer by `after_exchange`."""
        import torch

        dev = self._device_view(frame, level, kind)
        if not self._stage:
            return dev
        if self.slot(frame) >= 0:
            return dev.cpu()
        host = torch.empty(dev.shape, dtype=torch.uint8).pin_memory()
        self._pending.append((dev, host))
        return host

    def scratch(self, level, kind):
        import torch

        w, h = self.g.sizes[level]
        n = w * h * (self.g.S * 8 if kind == KIND_COLOR else self.g.S if kind == KIND_FG else self.g.D * 4)
        return torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))

    # ---- transports
    def attach_loopback(self, peers):
        arr = (C.c_void_p * len(peers))(*[p.h for p in peers])
        self._ck(derp.lib().derp_seq_attach_loopback(self.h, arr, len(peers)))
        self.transport = "loopback"

    def attach_rccl(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(derp.lib().derp_seq_attach_rccl(self.h, buf, 128))
        self.transport = "rccl"

    def attach_torch(self, dist, mode="p2p", stage_on_host=False):
        self._ck(derp.lib().derp_seq_attach_external(self.h))
        self._dist, self._mode, self._stage = dist, mode, stage_on_host
        self.transport = "torch" if mode == "p2p" else "broadcast"

    def selftest(self, words=4096):
        self._ck(derp.lib().derp_seq_selftest(self.h, words))

    def attach_best(self, dist, prefer=("rccl", "torch", "broadcast"), log=None):
        """Attach the first transport that every rank can use (agreement via all_reduce MIN)."""
        import torch

        def agree(ok):
            t = torch.tensor([1.0 if ok else 0.0], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)

        for name in prefer:
            ok, why = True, ""
            ident = [None]
            if name == "rccl":
                # ncclCommInitRank blocks until every rank has called it: first make sure, collectively, that every
                # rank can load librccl at all (a rank that cannot would leave the others waiting in the rendezvous)
                try:
                    rccl_unique_id()  # loads and binds the library; the id itself is discarded
                    loadable = True
                except Exception as e:  # noqa: BLE001
                    loadable, why = False, str(e)
                if not agree(loadable):
                    if log and self.rank == 0:
                        log("sequence: transport rccl unavailable (%s)" % (why or "a peer cannot load librccl"))
                    continue
                # every rank takes part in the broadcast even when rank 0 could not make an id
                if self.rank == 0:
                    try:
                        ident = [rccl_unique_id()]
                    except Exception as e:  # noqa: BLE001
                        why = str(e)
                dist.broadcast_object_list(ident, src=0)
            try:
                if name == "rccl":
                    if ident[0] is None:
                        raise RuntimeError(why or "rank 0 could not create an RCCL unique id")
                    self.attach_rccl(ident[0])
                    self.selftest()
                else:
                    self.attach_torch(dist, "p2p" if name == "torch" else "broadcast")
                    probe = torch.full((64,), float(self.rank), device="cuda")
                    got = torch.empty_like(probe)
                    nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
                    if name == "torch":
                        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, probe, nxt),
                                                         dist.P2POp(dist.irecv, got, prv)]):
                            w.wait()
                        torch.cuda.synchronize()
                        ok = bool((got == float(prv)).all().item())
            except Exception as e:  # noqa: BLE001
                ok, why = False, str(e)
            if agree(ok):
                return name
            if log and self.rank == 0:
                log("sequence: transport %s unavailable (%s)" % (name, why or "a peer failed"))
        raise RuntimeError("no usable transport between the ranks")

    # ---- schedule
    def exchange_inputs(self):
        if self.transport in ("torch", "broadcast"):
            transfers = plan(self.first, self.last, self.world, self.opt.time_radius, self.opt.partition)
            self.before_exchange()
            kinds = [KIND_COLOR] + ([KIND_FG] if self.opt.use_foreground_masks else [])
            for level in range(len(self.g.sizes)):
                # with an external transport this moves nothing; out of core it stages the frames other ranks read
                self._ck(derp.lib().derp_seq_exchange_inputs_level(self.h, level))
                for kind in kinds:
                    exchange(transfers, self.rank, lambda f: self.tensor(f, level, kind), self._dist, self._mode,
                             lambda: self.scratch(level, kind))
                    self.after_exchange()
        else:
            self._ck(derp.lib().derp_seq_exchange_inputs(self.h))

    def compute(self, level):
        self._ck(derp.lib().derp_seq_level_compute(self.h, level))

    def exchange_level(self, level):
        self._ck(derp.lib().derp_seq_level_exchange(self.h, level))

    def compute_frame(self, level, frame):
        self._ck(derp.lib().derp_seq_level_compute_frame(self.h, level, frame))

    def exchanged(self, level):
        """An external transport (torch.distributed) has delivered the halo frames' level."""
        self._ck(derp.lib().derp_seq_mark_exchanged(self.h, level))

    def filter(self, level):
        self._ck(derp.lib().derp_seq_level_filter(self.h, level))

    def filter_frame(self, level, frame):
        """One owned frame ahead of `filter` -> True if it could be filtered now (its window is complete)."""
        rc = derp.lib().derp_seq_level_filter_frame(self.h, level, frame)
        if rc == 2:
            return False
        self._ck(rc)
        return True

    def download_filtered(self, frame, level, d):
        import numpy as np

        w, h = self.g.sizes[level]
        out = np.zeros((h, w), dtype=np.float32)
        self._ck(derp.lib().derp_seq_download_filtered(self.h, frame, level, d, out.ctypes.data_as(C.c_void_p)))
        return out

    def before_exchange(self):
        self.g.synchronize()  # the level's kernels ran on the library's stream

    def after_exchange(self):
        import torch

        for dev, host in self._pending:
            dev.copy_(host)
        self._pending = []
        torch.cuda.synchronize()  # received tensors were produced on torch's / RCCL's streams

    def run(self, level_start=None, level_end=0):
        """All levels, coarse to fine, with whichever transport is attached."""
        level_start = len(self.g.sizes) - 1 if level_start is None else level_start
        if self.transport in ("local", "rccl"):
            self._ck(derp.lib().derp_seq_run(self.h, level_start, level_end))
        elif self.transport in ("torch", "broadcast"):
            self._received += run_schedule(self, range(level_start, level_end - 1, -1), self.first, self.last,
                                           self.rank, self.world, self.opt.time_radius, self.opt.partition,
                                           self._dist, self._mode)
        else:
            raise RuntimeError("transport %r needs the phases driven by the caller" % self.transport)

    def stats(self):
        a, b, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        self._ck(derp.lib().derp_seq_stats(self.h, C.byref(a), C.byref(b), C.byref(ms)))
        exposed = C.c_double()
        self._ck(derp.lib().derp_seq_exchange_exposed_ms(self.h, C.byref(exposed)))
        # exchange_ms: on the library's exchange stream; exchange_exposed_ms: what the compute stream waited of it (the rest
        # ran beside the filter of the frames whose windows are local)
        return dict(bytes_sent=a.value, bytes_received=b.value + self._received, exchange_ms=ms.value,
                    exchange_exposed_ms=exposed.value)

    def stats_reset(self):
        self._received = 0
        self._ck(derp.lib().derp_seq_stats_reset(self.h))


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    if derp.lib().derp_rccl_unique_id(buf, 128):
        raise RuntimeError("ncclGetUniqueId failed (librccl not loadable?)")
    return bytes(buf.raw)


def run_loopback(runners, level_start, level_end=0):
    """Several ranks emulated in one process on one GPU: phase by phase across all of them."""
    for r in runners:
        r.attach_loopback(runners)
    for r in runners:
        r.exchange_inputs()
    for level in range(level_start, level_end - 1, -1):
        for r in runners:
            r.compute(level)
        for r in runners:
            r.exchange_level(level)
        for r in runners:
            r.filter(level)
    for r in runners:
        r.g.synchronize()

