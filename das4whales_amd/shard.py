"""Channel-block sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Row-independent stages of the path -- dsp.bp_filt, detect.compute_cross_correlogram, picks --
shard by contiguous channel block with no communication; the filtered t-x matrix (or, much
cheaper, the correlogram / picks) is reassembled with ONE all-gather.  The f-k filter is a global
2-D transform: applying it per channel shard is a different filter, so it is not
offered here per shard.
"""
import torch
import torch.distributed as dist


def channel_block(nx, world_size, rank):
    """Contiguous balanced partition of nx channels: ranks < nx % world get one extra row."""
    base, extra = divmod(int(nx), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class _Works:
    """The handles of a batch of point-to-point transfers behind one wait() (what async_op=True returns for how="direct")."""

    def __init__(self, works):
        self._works = list(works)

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []


def all_gather_rows(local, nx_total, group=None, async_op=False, how="collective"):
    """Reassemble a row-sharded [rows_local, ...] tensor into [nx_total, ...] on every rank.

    how="collective" (default): a single all-gather (dist.all_gather_into_tensor).  Uneven shards are padded to the largest
    block for the collective and stripped afterwards.  RCCL picks the algorithm: a RING all-gather moves (N - 1) / N of the
    result over ONE xGMI link per step -- 8.4 GB into every GPU at N = 8 is >= 55 ms at ~153 GB/s per link.
    how="direct": every rank sends its block to each of the other N - 1 ranks itself, and receives theirs straight into the
    result's row ranges -- N - 1 grouped isend / irecv pairs (dist.batch_isend_irecv), one per peer, so that all seven
    point-to-point links of a GPU carry 1 / 7 of the traffic each (>= 7.8 ms for the same bytes); no padding for uneven shards.
    async_op=True returns (out, work): the transfers run behind whatever the caller launches next -- e.g. the matched filter of
    the local rows -- until work.wait() (collective form: even shards only)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    blocks = [channel_block(nx_total, world, r) for r in range(world)]
    if local.shape[0] != blocks[rank][1] - blocks[rank][0]:
        raise ValueError("local block has %d rows, expected %d" % (local.shape[0], blocks[rank][1] - blocks[rank][0]))
    if how not in ("collective", "direct"):
        raise ValueError("how must be 'collective' or 'direct', not %r" % (how,))
    rmax = max(b[1] - b[0] for b in blocks)
    tail = tuple(local.shape[1:])
    local = local.contiguous()
    if how == "direct":
        out = torch.empty((nx_total,) + tail, dtype=local.dtype, device=local.device)
        out[blocks[rank][0]:blocks[rank][1]].copy_(local)
        ops = []
        for step in range(1, world):                     # peer rank + step sends to us what we send to rank - step ... all at once
            dst, src = (rank + step) % world, (rank - step) % world
            gdst = dist.get_global_rank(group, dst) if group is not None else dst
            gsrc = dist.get_global_rank(group, src) if group is not None else src
            if blocks[rank][1] > blocks[rank][0]:
                ops.append(dist.P2POp(dist.isend, local, gdst, group))
            if blocks[src][1] > blocks[src][0]:
                ops.append(dist.P2POp(dist.irecv, out[blocks[src][0]:blocks[src][1]], gsrc, group))
        work = _Works(dist.batch_isend_irecv(ops) if ops else [])
        if async_op:
            return out, work
        work.wait()
        return out
    if nx_total % world == 0:
        out = torch.empty((nx_total,) + tail, dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError("async_op with how='collective' needs even channel blocks")
    padded = torch.zeros((rmax,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * rmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * rmax: r * rmax + (b[1] - b[0])] for r, b in enumerate(blocks)], dim=0)


def all_gather_picks(packed_local, row_begin, group=None):
    """Reassemble the picks of a channel-sharded block on every rank: `packed_local` is this rank's 2 x K_r table
    (row 0 = LOCAL channel, row 1 = time index: detect.PickRows.packed / detect.convert_pick_times) and `row_begin` the
    global index of its first channel.  Returns the global 2 x K int64 table in channel order -- what
    detect.convert_pick_times gives for the whole block.  Two small collectives (the counts, then the padded tables): a few
    MB instead of the 9.6 GB all-gather of the filtered t-x matrix ("prefer gathering detections")."""
    world = dist.get_world_size(group)
    dev_ = packed_local.device
    k_loc = torch.tensor([packed_local.shape[1]], dtype=torch.int64, device=dev_)
    counts = torch.empty(world, dtype=torch.int64, device=dev_)
    dist.all_gather_into_tensor(counts, k_loc, group=group)
    counts_h = [int(v) for v in counts.cpu()]
    kmax = max(counts_h) if counts_h else 0
    if kmax == 0:
        return torch.zeros((2, 0), dtype=torch.int64, device=dev_)
    pad = torch.zeros((2, kmax), dtype=torch.int64, device=dev_)
    pad[:, :packed_local.shape[1]] = packed_local.to(torch.int64)
    pad[0, :packed_local.shape[1]] += int(row_begin)
    buf = torch.empty((world, 2, kmax), dtype=torch.int64, device=dev_)
    dist.all_gather_into_tensor(buf, pad.unsqueeze(0), group=group)
    return torch.cat([buf[r, :, :counts_h[r]] for r in range(world)], dim=1)


def map_channel_blocks(fn, x_full, gather=True, group=None):
    """Apply a row-independent operator to this rank's channel block of `x_full` ([nx, ns], present
    on every rank or memory-mapped) and optionally all-gather the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    a, b = channel_block(x_full.shape[0], world, rank)
    y = fn(x_full[a:b])
    return all_gather_rows(y, x_full.shape[0], group=group) if gather else y


# ---------------------------------------------------------------------------------------------
# exact f-k filter of ONE block sharded by channel block (pencil decomposition, two all-to-alls)
# ---------------------------------------------------------------------------------------------
def _native():
    from ._lib import lib, check
    return lib, check


def _sptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


class ShardedFkPlan:
    """Per-rank plan of the distributed f-k filter (include/d4w.h, d4w_fkd_*): this rank's channel
    block, the owner of every time-axis sub-row q1, and the folded mask of the sub-rows it owns.
    `native` = (lib, check) lets the CPU tests drive the same code with the emulator build."""

    def __init__(self, nx, ns, group=None, native=None):
        import ctypes
        self._ct = ctypes
        self.lib, self.check = native if native is not None else _native()
        self._emulated = native is not None
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._h = ctypes.c_void_p()
        force_generic = False if native is not None else self._specialise(int(nx), int(ns))
        import os
        old_env = os.environ.get("D4W_FKD_GENERIC")
        if force_generic:
            os.environ["D4W_FKD_GENERIC"] = "1"
        try:
            self.check(self.lib.d4w_fkd_plan_create(int(nx), int(ns), self.world, self.rank, ctypes.byref(self._h)))
        finally:
            if force_generic:
                if old_env is None:
                    del os.environ["D4W_FKD_GENERIC"]
                else:
                    os.environ["D4W_FKD_GENERIC"] = old_env
        info = (ctypes.c_int * 12)()
        self.check(self.lib.d4w_fkd_plan_info(self._h, info))
        (self.nx, self.ns, _, _, self.row_begin, self.row_end, self.N1, self.N2, self.nq, self.C1, _, packed) = list(info)
        self.packed = bool(packed)          # shape-specialised kernels + exchange buffers that need no repacking
        own = (ctypes.c_int * self.N1)()
        self.check(self.lib.d4w_fkd_plan_q1_owner(self._h, own))
        owner = torch.tensor(list(own), dtype=torch.int64)
        self.qidx = [torch.nonzero(owner == s).flatten() for s in range(self.world)]
        self.blocks = [channel_block(self.nx, self.world, r) for r in range(self.world)]
        assert self.blocks[self.rank] == (self.row_begin, self.row_end)

    def sub_rows_per_rank(self):
        """Time-axis sub-rows (of N1) each rank transforms in the channel phase."""
        return [int(q.numel()) for q in self.qidx]

    def channel_phase_balance(self):
        """Mean over busiest rank of the channel-phase work: N1 / (world * max sub-rows of a rank).  A sub-row and its
        Hermitian partner stay together, so 25 sub-rows over 8 ranks cannot do better than 4 on the busiest (0.78)."""
        return self.N1 / float(self.world * max(1, max(self.sub_rows_per_rank())))

    def _specialise(self, nx, ns):
        """The ranks must agree on the plan type -- the packed plan (shape-specialised kernels) and the generic plan exchange
        different layouts -- whatever their environments and kernel caches say.  A new large shape first gets its kernels as
        in dsp.get_fk_plan (das4whales_amd/fkjit.py: ~15 s once, cached on disk): rank 0 compiles, the others pick the
        cached object up.  Then EVERY rank reports whether the shape is specialised in its process and the outcome is
        min-reduced over the group (always, also for small shapes and with D4W_FK_JIT=0: each rank could otherwise decide
        from its own cache and skip the collective the others wait in).  Returns True when this rank must hold back to the
        generic plan because some rank has no specialised kernels."""
        import os
        want_jit = os.environ.get("D4W_FK_JIT", "1") != "0" and nx * ns >= (1 << 24)
        if self.world > 1:                                   # the decision to compile must be the same everywhere, too
            w = torch.tensor([int(want_jit)], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(w, op=dist.ReduceOp.MIN, group=self.group)
            want_jit = bool(int(w.item()))

        def attempt():
            try:
                from . import fkjit
                fkjit.compile_fk_shape(nx, ns)
            except Exception:
                pass
        if want_jit:
            if self.rank == 0:
                attempt()
            if self.world > 1:
                dist.barrier(self.group)
                if self.rank != 0:
                    attempt()
        if self.world == 1:
            return False
        mine = int(bool(self.lib.d4w_fk_shape_is_specialised(int(nx), int(ns))))
        flag = torch.tensor([mine], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(mine) and int(flag.item()) == 0

    def set_mask(self, mask):
        """mask: dense float32 [nx, ns] tensor on this rank's device, fftshift-ed grid (what the
        reference designers return; every rank designs or loads the same mask), or the closed-form
        dsp.DesignedMask that fk_filter_design / hybrid_filter_design / hybrid_ninf_filter_design return --
        then every rank evaluates only the gains of the sub-rows it owns and no dense mask exists anywhere."""
        if tuple(mask.shape) != (self.nx, self.ns):
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) %s"
                             % (self.nx, self.ns, tuple(mask.shape)))
        if hasattr(mask, "hrow_on"):
            ct = self._ct
            device = torch.device("cpu") if self._emulated else torch.device("cuda", torch.cuda.current_device())
            h = mask.hrow_on(device)
            p8 = (ct.c_double * 8)(*mask.params)
            self.check(self.lib.d4w_fkd_set_mask_design_f32(
                self._h, ct.c_int(mask.mode), ct.c_double(mask.k_spacing), ct.c_double(mask.t_spacing), p8, ct.c_int(mask.i0),
                ct.c_int(mask.i1), ct.c_void_p(h.data_ptr()) if h is not None else None,
                ct.c_void_p(None if self._emulated else torch.cuda.current_stream(device).cuda_stream)))
            if not self._emulated:
                torch.cuda.current_stream(device).synchronize()
            return
        m = mask.to(torch.float32).contiguous()
        self.check(self.lib.d4w_fkd_set_mask_dense_f32(self._h, m.data_ptr(), _sptr(m)))
        if m.is_cuda:
            torch.cuda.current_stream(m.device).synchronize()

    # elements per collective call: RCCL / gloo counts are 32-bit safe well below this, and the
    # chunking also bounds the size of the packed send / receive staging buffers
    MAX_CALL_ELEMS = 1 << 29

    def _chunk_rows(self, r, j, nch):
        """Global row range of chunk j (of nch) of rank r's channel block."""
        a, b = self.blocks[r]
        n = b - a
        return a + (n * j) // nch, a + (n * (j + 1)) // nch

    def _scratch(self, name, numel, device):
        """Persistent staging buffers (allocated once per plan: multi-GB temporaries churn the allocator)."""
        pool = self.__dict__.setdefault("_pool", {})
        t = pool.get(name)
        if t is None or t.numel() < numel or t.device != device:
            t = pool[name] = torch.empty(int(numel), dtype=torch.float32, device=device)
        return t[:numel]

    def _all_to_all(self, recv, send, out_splits, in_splits):
        if self.world == 1:
            recv.copy_(send)
        else:
            dist.all_to_all_single(recv, send, out_splits, in_splits, group=self.group)

    # Row chunks of the packed exchanges (transfer of chunk i overlaps the time-axis transform of chunk i + 1).  None = from
    # the bytes a rank sends to one peer per exchange: messages of >= 16 MiB (0.1 ms on one 153-GB/s xGMI link -- below that the
    # per-message cost of a grouped isend / irecv shows), at most 8 chunks; every rank derives the same count from the plan's
    # shape (20 000 x 120 000 over 8 ranks: 131 MB per peer -> 8 chunks; over 2 ranks: 2.4 GB -> 8).  An integer pins it.
    CHUNKS = None

    def exchange_chunks(self):
        """Number of row chunks per exchange (the same on every rank)."""
        if self.CHUNKS is not None:
            want = int(self.CHUNKS)
        else:
            per_peer = (self.nx / max(self.world, 1)) * self.ns * 4.0 / max(self.world, 1)
            want = int(min(8, max(1, per_peer // (16 << 20))))
        return max(1, min([want] + [-(-(b - a) // self.C1) for a, b in self.blocks]))

    def _chunks(self, r, nch):
        """Local row ranges [l0, l1) of rank r's nch chunks: boundaries at multiples of C1 (the time-phase tile)."""
        n = self.blocks[r][1] - self.blocks[r][0]
        grp = -(-n // self.C1)
        cut = [min(n, self.C1 * ((grp * j) // nch)) for j in range(nch)] + [n]
        return [(cut[j], cut[j + 1]) for j in range(nch)]

    def _exchange(self, send_views, recv_views):
        """One grouped exchange: send_views[s] goes to rank s, recv_views[r] arrives from rank r (contiguous float32
        views; the own piece is a device copy).  Returns the outstanding work handles."""
        recv_views[self.rank].copy_(send_views[self.rank])
        ops = []
        for k in range(1, self.world):                      # pairwise schedule: every link busy, no hot receiver
            to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
            if send_views[to].numel():
                ops.append(dist.P2POp(dist.isend, send_views[to], to, group=self.group))
            if recv_views[frm].numel():
                ops.append(dist.P2POp(dist.irecv, recv_views[frm], frm, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _mark(self, label):
        """Stage boundary for bench.py's per-stage times (no-op unless `self.marks` is a list and the data is on a GPU)."""
        marks = getattr(self, "marks", None)
        if marks is not None and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((label, e))

    def _apply_packed(self, x_loc, taper, stats=False):
        """Packed plan (include/d4w.h): 7 block passes of the shape-specialised kernels per rank, the exchange
        buffers are written / read in place by the time-phase kernels (no index packing), and the transfers run in
        row chunks behind the kernels of the neighbouring chunks."""
        nxl = self.row_end - self.row_begin
        per = self.N2 * 2
        dev_ = x_loc.device
        nqs = [len(q) for q in self.qidx]
        send = self._scratch("send", nxl * self.N1 * per, dev_)
        boff = [0]
        for s in range(self.world):
            boff.append(boff[-1] + nxl * nqs[s] * per)
        mean = mx = None
        if stats:
            mean = torch.zeros(nxl, dtype=torch.float64, device=dev_)
            mx = torch.zeros(nxl, dtype=torch.float32, device=dev_)

        def time_inv(buf, y, l0, l1):
            if stats:
                self.check(self.lib.d4w_fkd_time_inv_packed_rows_stats_f32(self._h, buf.data_ptr(), y.data_ptr(), l0, l1,
                                                                             mean.data_ptr(), mx.data_ptr(), _sptr(y)))
            else:
                self.check(self.lib.d4w_fkd_time_inv_packed_rows_f32(self._h, buf.data_ptr(), y.data_ptr(), l0, l1, _sptr(y)))
        if self.world == 1:                                   # one destination: the send buffer IS the slab
            self.check(self.lib.d4w_fkd_time_fwd_packed_f32(self._h, x_loc.data_ptr(), send.data_ptr(), int(bool(taper)), _sptr(send)))
            self.check(self.lib.d4w_fkd_chan_apply_f32(self._h, send.data_ptr(), _sptr(send)))
            y = torch.empty((nxl, self.ns), dtype=torch.float32, device=dev_)
            time_inv(send, y, 0, nxl)
            return (y, mean, mx) if stats else y
        slab = self._scratch("slab", self.nx * self.nq * per, dev_)
        # the same chunk count on every rank (uneven channel blocks differ by a row: a count derived from this rank's own
        # block could differ between ranks, and the grouped send / recv lists would then not match)
        nch = self.exchange_chunks()
        chunks = [self._chunks(r, nch) for r in range(self.world)]
        works = []
        self._mark("start")
        for j in range(nch):
            l0, l1 = chunks[self.rank][j]
            self.check(self.lib.d4w_fkd_time_fwd_packed_rows_f32(self._h, x_loc.data_ptr(), send.data_ptr(), int(bool(taper)),
                                                                   l0, l1, _sptr(send)))
            sv = [send[boff[s] + l0 * nqs[s] * per: boff[s] + l1 * nqs[s] * per] for s in range(self.world)]
            rv = []
            for r in range(self.world):
                a = self.blocks[r][0]
                r0, r1 = chunks[r][j]
                rv.append(slab[(a + r0) * self.nq * per:(a + r1) * self.nq * per])
            works += self._exchange(sv, rv)
        for w in works:
            w.wait()
        self._mark("time_fwd+exchange")
        self.check(self.lib.d4w_fkd_chan_apply_f32(self._h, slab.data_ptr() if self.nq else None, _sptr(slab)))
        self._mark("channel_phase")
        y = torch.empty((nxl, self.ns), dtype=torch.float32, device=dev_)
        pending = []
        for j in range(nch):                                  # all transfers are queued; chunk j is transformed as it lands
            l0, l1 = chunks[self.rank][j]
            rv = [send[boff[s] + l0 * nqs[s] * per: boff[s] + l1 * nqs[s] * per] for s in range(self.world)]
            sv = []
            for r in range(self.world):
                a = self.blocks[r][0]
                r0, r1 = chunks[r][j]
                sv.append(slab[(a + r0) * self.nq * per:(a + r1) * self.nq * per])
            pending.append(self._exchange(sv, rv))
        for j in range(nch):
            for w in pending[j]:
                w.wait()
            l0, l1 = chunks[self.rank][j]
            time_inv(send, y, l0, l1)
        self._mark("exchange+time_inv")
        return (y, mean, mx) if stats else y

    def apply(self, x_loc, taper=False, stats=False):
        """x_loc: float32 [rows of this rank, ns] -> filtered rows of this rank (same shape).  stats=True (packed plans)
        also returns (mean, max|.|) of every filtered local row, formed in the last pass's epilogue."""
        nxl = self.row_end - self.row_begin
        if tuple(x_loc.shape) != (nxl, self.ns):
            raise ValueError("local block has shape %s, expected (%d, %d)" % (tuple(x_loc.shape), nxl, self.ns))
        x_loc = x_loc.to(torch.float32).contiguous()
        if self.packed:
            return self._apply_packed(x_loc, taper, stats)
        if stats:
            raise ValueError("row statistics come with the packed plan only")
        per = self.N2 * 2                                        # floats per sub-row
        z = torch.empty((nxl, self.N1, per), dtype=torch.float32, device=x_loc.device)     # returned to the caller
        self.check(self.lib.d4w_fkd_time_fwd_f32(self._h, x_loc.data_ptr(), z.data_ptr(), int(bool(taper)), _sptr(z)))
        qidx = [q.to(z.device) for q in self.qidx]
        nql = [len(q) for q in self.qidx]
        biggest = max(b - a for a, b in self.blocks) * self.N1 * per
        nch = max(1, -(-biggest // self.MAX_CALL_ELEMS))          # same on every rank
        slab = self._scratch("slab", self.nx * self.nq * per, z.device).view(self.nx, self.nq * per)   # [nx][nq][N2] complex
        # exchange 1: sub-row q1 of every local channel -> rank owner[q1], in row chunks
        for j in range(nch):
            g0, g1 = self._chunk_rows(self.rank, j, nch)
            l0, l1 = g0 - self.row_begin, g1 - self.row_begin
            to_peer = [(l1 - l0) * nql[s] * per for s in range(self.world)]
            send = self._scratch("send", sum(to_peer), z.device)
            off = 0
            for s in range(self.world):
                if to_peer[s]:
                    torch.index_select(z[l0:l1], 1, qidx[s], out=send[off:off + to_peer[s]].view(l1 - l0, nql[s], per))
                off += to_peer[s]
            rows = [self._chunk_rows(r, j, nch) for r in range(self.world)]
            from_peer = [(b - a) * self.nq * per for a, b in rows]
            recv = slab.view(-1) if nch == 1 else self._scratch("recv", sum(from_peer), z.device)
            self._all_to_all(recv, send, from_peer, to_peer)
            if nch > 1:
                off = 0
                for (a, b), n in zip(rows, from_peer):
                    slab[a:b] = recv[off:off + n].view(b - a, self.nq * per)
                    off += n
        self.check(self.lib.d4w_fkd_chan_apply_f32(self._h, slab.data_ptr() if self.nq else None, _sptr(slab)))
        # exchange 2: the exact reverse
        for j in range(nch):
            g0, g1 = self._chunk_rows(self.rank, j, nch)
            l0, l1 = g0 - self.row_begin, g1 - self.row_begin
            rows = [self._chunk_rows(r, j, nch) for r in range(self.world)]
            to_peer = [(b - a) * self.nq * per for a, b in rows]
            from_peer = [(l1 - l0) * nql[s] * per for s in range(self.world)]
            if nch == 1:
                send = slab.view(-1)
            else:
                send = self._scratch("recv", sum(to_peer), z.device)
                off = 0
                for (a, b), n in zip(rows, to_peer):
                    send[off:off + n].view(b - a, self.nq * per).copy_(slab[a:b])
                    off += n
            back = self._scratch("send", sum(from_peer), z.device)
            self._all_to_all(back, send, from_peer, to_peer)
            off = 0
            for s in range(self.world):
                n = from_peer[s]
                if n:
                    z[l0:l1].index_copy_(1, qidx[s], back[off:off + n].view(l1 - l0, nql[s], per))
                off += n
        self.check(self.lib.d4w_fkd_time_inv_f32(self._h, z.data_ptr(), _sptr(z)))
        return z.view(nxl, self.ns)

    def __del__(self):
        try:
            if self._h:
                self.lib.d4w_fkd_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def fk_filter_sharded(x_loc, fk_filter_matrix, nx_total, tapering=False, gather=False, group=None, plan=None):
    """dsp.fk_filter_filt (reference dsp.py:725-756) of ONE [nx_total, ns] block whose rows are
    sharded by contiguous channel block (channel_block) over the ranks of `group`.  Returns this
    rank's filtered rows, or the whole filtered matrix on every rank after a single all-gather
    when gather=True (north star: RCCL all-gather of the t-x output)."""
    if plan is None:
        plan = ShardedFkPlan(nx_total, x_loc.shape[1], group=group)
        if hasattr(fk_filter_matrix, "hrow_on"):          # closed-form design: straight into the plan
            plan.set_mask(fk_filter_matrix)
        else:
            m = fk_filter_matrix.tensor if hasattr(fk_filter_matrix, "tensor") else fk_filter_matrix
            if not isinstance(m, torch.Tensor):
                import numpy as np
                m = torch.from_numpy(np.ascontiguousarray(np.asarray(m.todense() if hasattr(m, "todense") else m),
                                                          dtype=np.float32))
            plan.set_mask(m.to(x_loc.device))
    y = plan.apply(x_loc, taper=tapering)
    return all_gather_rows(y, nx_total, group=group) if gather else y
