"""z-slab domain decomposition for multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

A volume [z][y][x] of nz_global slices is cut into `world` contiguous slabs along z (the slowest axis, so every
slab face is one contiguous run of ny*nx floats).  Each rank stores its slab plus `halo` slices of its
neighbours on the interior sides; the two ends of the volume carry no halo, so the array edge of an end rank IS
the domain boundary and the kernels' boundary rules (OOB constants, edge replication, one-sided differences)
apply unchanged.  After every iteration the dynamic fields' boundary slices are exchanged point-to-point with
the (at most two) neighbours, and the iteration record (max update, energies) is all-reduced.

The reference has no distributed code at all; this module is new design, see DESIGN.md section 6.
"""
import ctypes
import os
import warnings

import torch
import torch.distributed as dist


from . import _lib

# lsf_iteration_record (include/lsf_hip.h): LSF_RECORD_SLOTS partial slots, 4 KiB (512 int64 words) apart; word 0 of a
# slot = packed max, words 1..3 = energies.  The layout is the binding's (derived from the ctypes structures there).
RECORD_SLOTS, SLOT_WORDS = _lib.RECORD_SLOTS, _lib.SLOT_WORDS


def _slot_view(records):
    return records.view(records.shape[0], RECORD_SLOTS, SLOT_WORDS)


class SlabLayout:
    """`world` contiguous slabs of a volume along `axis` (0 = z, the default: a slab and its faces are contiguous runs;
    1 = y: for volumes whose narrow band is a sheet across z -- a depth frame's surface, tsdf/generation.py:356-437 --,
    which z-slabs would leave to one or two ranks).  Generic names: `begin` / `end` = owned range inside the local array,
    `n_local` = local extent incl. halos, `global_offset` = global index of local index 0, `g0` / `g1` = owned range in
    global indices.  The z-named attributes are the same numbers and are only meaningful for axis 0."""

    def __init__(self, nz_global, rank=0, world=1, halo=0, axis=0):
        if axis not in (0, 1):
            raise ValueError("slabs are cut along z (axis 0) or y (axis 1)")
        if nz_global % world != 0:
            raise ValueError("extent %d is not divisible by the number of slabs %d" % (nz_global, world))
        per = nz_global // world
        if world > 1 and halo > per:
            raise ValueError("halo %d wider than a slab of %d slices" % (halo, per))
        self.axis = axis
        self.nz_global, self.rank, self.world, self.halo = nz_global, rank, world, halo
        self.z0, self.z1 = rank * per, (rank + 1) * per
        self.halo_lo = halo if rank > 0 else 0
        self.halo_hi = halo if rank < world - 1 else 0
        self.nz_local = per + self.halo_lo + self.halo_hi
        self.z_begin, self.z_end = self.halo_lo, self.halo_lo + per
        self.z_global_offset = self.z0 - self.halo_lo
        self.n_global, self.g0, self.g1 = self.nz_global, self.z0, self.z1
        self.n_local, self.begin, self.end, self.global_offset = self.nz_local, self.z_begin, self.z_end, \
            self.z_global_offset

    def local_slice(self):
        """slice of the GLOBAL partition axis held locally (owned slab + halos)"""
        return slice(self.z0 - self.halo_lo, self.z1 + self.halo_hi)

    def owned_local(self):
        return slice(self.z_begin, self.z_end)

    def cut(self, volume):
        """the local part (owned slab + halos) of a whole [z][y][x] volume"""
        return (volume[self.local_slice()] if self.axis == 0 else volume[:, self.local_slice()]).contiguous()

    def owned_of(self, local):
        """the owned part of a local [z][y][x] (or [z][y][x][c]) array"""
        return local[self.owned_local()] if self.axis == 0 else local[:, self.owned_local()]


class SlabComm:
    """halo exchange + record reduction for one slab layout.  `group` = a torch.distributed process group
    (None = default group).  With world == 1 every method is a no-op."""

    def __init__(self, layout, group=None):
        self.layout = layout
        self.group = group
        # RCCL ("nccl") moves device buffers directly over xGMI.  gloo has no device point-to-point: device tensors
        # are staged through host memory then (used by the 2-ranks-on-one-GPU test of the whole N > 1 flow).
        self.stage_through_host = layout.world > 1 and dist.is_initialized() and dist.get_backend(group) == "gloo"

    @property
    def active(self):
        return self.layout.world > 1

    # ---- native transport: the library's own RCCL communicator (lsf_slab.hip) -------------------------------------
    def native_identity(self):
        """(rank, world, lower neighbour, upper neighbour) inside the native communicator; -1 = no neighbour"""
        L = self.layout
        return L.rank, L.world, (L.rank - 1 if L.rank > 0 else -1), (L.rank + 1 if L.rank < L.world - 1 else -1)

    def native(self):
        """handle of the library-side communicator (created on first use: rank 0 draws the RCCL unique id, the bytes
        travel over torch.distributed), or None when the transport is torch.distributed -- CPU / gloo runs,
        LSF_SLAB_TRANSPORT=torch, or RCCL could not be bound"""
        if hasattr(self, "_native"):
            return self._native
        self._native = None
        if not self.active or self.stage_through_host or not dist.is_initialized() \
                or os.environ.get("LSF_SLAB_TRANSPORT", "rccl") == "torch":
            return None
        from . import _lib
        rank, world, _, _ = self.native_identity()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = path.encode() if os.path.exists(path) else None
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        # every rank draws an id (only rank 0's is used): that binds RCCL, and the ranks agree on the outcome BEFORE
        # anyone enters the collective communicator set-up -- a rank that cannot bind must not leave the others waiting
        host = (ctypes.c_uint8 * 128)()
        bound = torch.tensor([int(_lib.lib.lsf_slab_unique_id(path, ctypes.cast(host, ctypes.c_void_p)) == 0)],
                             dtype=torch.int32, device="cuda")
        if dist.get_world_size(self.group) > 1:
            dist.all_reduce(bound, op=dist.ReduceOp.MIN, group=self.group)
        if int(bound.item()) == 0:
            warnings.warn("native RCCL slab transport unavailable (RCCL could not be bound on every rank); using "
                          "torch.distributed point-to-point")
            return None
        try:
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
            if dist.get_world_size(self.group) > 1:
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast(uid, src=src, group=self.group)
            host = (ctypes.c_uint8 * 128)(*uid.cpu().tolist())
            handle = ctypes.c_void_p()
            _lib.check(_lib.lib.lsf_slab_comm_create(path, ctypes.cast(host, ctypes.c_void_p), rank, world,
                                                     ctypes.byref(handle)), "lsf_slab_comm_create")
            self._native = handle
        except Exception as exc:  # noqa: BLE001 -- any failure here only costs speed: the torch transport still works
            warnings.warn("native RCCL slab transport unavailable (%s); using torch.distributed point-to-point" % exc)
            self._native = None
        return self._native

    def native_info(self):
        """(rank, number of ranks) as RCCL reports them for the native communicator (ncclCommUserRank, ncclCommCount), or
        None on the torch.distributed transport"""
        handle = self.native()
        if handle is None:
            return None
        rank, count = ctypes.c_int32(-1), ctypes.c_int32(-1)
        _lib.check(_lib.lib.lsf_slab_comm_info(handle, ctypes.byref(rank), ctypes.byref(count)), "lsf_slab_comm_info")
        return int(rank.value), int(count.value)

    def close(self):
        """release the native communicator (before torch.distributed.destroy_process_group)"""
        handle = getattr(self, "_native", None)
        if handle is not None:
            from . import _lib
            _lib.lib.lsf_slab_comm_destroy(handle)
        self._native = None

    def _z_view(self, t, a, b):
        # scalar field [z,y,x] or planar vector field [c,z,y,x]; slices [a, b) along the layout's partition axis
        if self.layout.axis == 0:
            return t[a:b] if t.dim() == 3 else t[:, a:b]
        return t[:, a:b] if t.dim() == 3 else t[:, :, a:b]

    def _staging(self, key, shape, like):
        """persistent send / receive staging buffers (allocated once per distinct message shape)"""
        if not hasattr(self, "_buffers"):
            self._buffers = {}
        if key not in self._buffers:
            device = "cpu" if (self.stage_through_host and like.is_cuda) else like.device
            self._buffers[key] = (torch.empty(shape, dtype=like.dtype, device=device),
                                  torch.empty(shape, dtype=like.dtype, device=device))
        return self._buffers[key]

    def exchange_halos(self, tensors, width=None):
        """fill the halo slices of every tensor from the neighbours' owned boundary slices.  All tensors travel in
        ONE message per neighbour and direction (scalar fields [z,y,x] count one channel, planar vector fields
        [c,z,y,x] c channels), through persistent staging buffers."""
        L = self.layout
        if not self.active:
            return
        h = L.halo if width is None else width
        if h == 0:
            return
        if h > L.halo:
            raise ValueError("requested halo width %d exceeds the layout's halo %d" % (h, L.halo))
        channels = [1 if t.dim() == 3 else t.shape[0] for t in tensors]
        probe = self._z_view(tensors[0], 0, h)
        plane = tuple(probe.shape[-3:]) if L.axis == 1 else tuple(tensors[0].shape[-2:])  # one message row per channel
        shape = (sum(channels),) + (plane if L.axis == 1 else (h,) + plane)
        ops, unpack = [], []
        # (neighbour rank, owned slices to send, halo slices to fill)
        sides = []
        if L.rank > 0:
            sides.append(("lo", L.rank - 1, (L.z_begin, L.z_begin + h), (L.z_begin - h, L.z_begin)))
        if L.rank < L.world - 1:
            sides.append(("hi", L.rank + 1, (L.z_end - h, L.z_end), (L.z_end, L.z_end + h)))
        for tag, peer, (sa, sb), (ra, rb) in sides:
            send, recv = self._staging((tag, shape, tensors[0].dtype), shape, tensors[0])
            k = 0
            for t, c in zip(tensors, channels):
                src = self._z_view(t, sa, sb)
                send[k:k + c].copy_(src if t.dim() == 4 else src.unsqueeze(0))
                k += c
            ops.append(dist.P2POp(dist.isend, send, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, recv, peer, self.group))
            unpack.append((recv, ra, rb))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for recv, ra, rb in unpack:
            k = 0
            for t, c in zip(tensors, channels):
                dst = self._z_view(t, ra, rb)
                dst.copy_(recv[k:k + c] if t.dim() == 4 else recv[k])
                k += c

    def exchange_live_and_warp(self, live, warp_planar):
        """the per-iteration exchange of the Slavcheva engine: one device kernel packs the boundary slices of the live
        field and the warp for BOTH neighbours, one batch of point-to-point operations moves them, one kernel unpacks
        into the halos (device tensors, width = the layout's halo)"""
        L = self.layout
        if not self.active or L.halo == 0:
            return
        if self.stage_through_host or not live.is_cuda:
            return self.exchange_halos([live, warp_planar])
        from . import device as dev
        h = L.halo
        shape = (1 + warp_planar.shape[0], h) + tuple(live.shape[-2:])
        lo = self._staging(("lo", shape, live.dtype), shape, live) if L.rank > 0 else (None, None)
        hi = self._staging(("hi", shape, live.dtype), shape, live) if L.rank < L.world - 1 else (None, None)
        dev.halo_copy(live, warp_planar, lo[0], hi[0], h, L.z_begin, L.z_end - h, unpack=False)
        ops = []
        if lo[0] is not None:
            ops += [dist.P2POp(dist.isend, lo[0], L.rank - 1, self.group),
                    dist.P2POp(dist.irecv, lo[1], L.rank - 1, self.group)]
        if hi[0] is not None:
            ops += [dist.P2POp(dist.isend, hi[0], L.rank + 1, self.group),
                    dist.P2POp(dist.irecv, hi[1], L.rank + 1, self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        dev.halo_copy(live, warp_planar, lo[1], hi[1], h, L.z_begin - h if lo[1] is not None else 0, L.z_end,
                      unpack=True)

    def exchange_state(self, state):
        """the per-iteration exchange of the fused Slavcheva path: the float4 state [z][y][x][4] keeps the live field
        and the warp together, so a halo is ONE contiguous run per neighbour and direction -- sent from and received
        into the tensor itself (RCCL point-to-point), no pack / unpack kernels, no staging buffers"""
        L = self.layout
        if not self.active or L.halo == 0:
            return
        if self.stage_through_host or not state.is_cuda or L.axis != 0:
            # y-slabs: a face is nz runs of halo * nx float4 -- packed into / out of persistent staging buffers
            return self.exchange_halos([state.view(state.shape[0], state.shape[1], -1)])
        h = L.halo
        _, _, lo, hi = self.native_identity()
        # order as in lsf_slab.hip: irrelevant between distinct peers, and the right pairing (lower boundary -> upper
        # halo) when a rank is its own neighbour in the one-GPU loop-back of a z-periodic stack
        ops = []
        if lo >= 0:
            ops.append(dist.P2POp(dist.isend, state[L.z_begin:L.z_begin + h], lo, self.group))
        if hi >= 0:
            ops.append(dist.P2POp(dist.irecv, state[L.z_end:L.z_end + h], hi, self.group))
            ops.append(dist.P2POp(dist.isend, state[L.z_end - h:L.z_end], hi, self.group))
        if lo >= 0:
            ops.append(dist.P2POp(dist.irecv, state[L.z_begin - h:L.z_begin], lo, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def all_gather_owned(self, local):
        """the OWNED slices of every rank's local array, concatenated along z: the whole level on every rank (static
        operands of the hierarchical optimizer whose gather outgrows the halo).  One collective."""
        own = local[self.layout.owned_local()].contiguous()
        if not self.active:
            return own
        world = dist.get_world_size(self.group)
        mine = own.cpu() if (self.stage_through_host and own.is_cuda) else own
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=self.group)
        return torch.cat(parts, 0).to(local.device)

    def gather_rows(self, value):
        """every rank's copy of a small float64 device tensor, as a list of host arrays in rank order"""
        if not self.active:
            return [value.cpu().numpy()]
        world = dist.get_world_size(self.group)
        mine = value.cpu() if (self.stage_through_host and value.is_cuda) else value
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine.contiguous(), group=self.group)
        return [r.cpu().numpy() for r in rows]

    def reduce_scalar_max(self, value):
        """in-place MAX all-reduce of a small float tensor (slab guards)"""
        if not self.active:
            return
        if self.stage_through_host and value.is_cuda:
            host = value.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            value.copy_(host)
        else:
            dist.all_reduce(value, op=dist.ReduceOp.MAX, group=self.group)

    def reduce_max(self, records, index):
        """MAX all-reduce of the packed maxima of ONE record (needed before a gated iteration can test it); a record
        is 8 partial slots (lsf_iteration_record), reduced slot by slot"""
        if not self.active:
            return
        view = _slot_view(records)[index, :, 0]
        mx = view.contiguous()
        if self.stage_through_host and records.is_cuda:
            mx = mx.cpu()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        view.copy_(mx.to(records.device))

    def gather_records(self, records, first, last):
        """records [first, last) of EVERY rank on the host: numpy int64 [n, world * slots, 4] -- the ranks' partial slots
        side by side, which is all device.decode_records needs (max of the packed maxima, sum of the energies).  One
        collective and one host read where reduce_records + records_to_host take two collectives; for runs whose
        records no device-side gate reads (fixed iteration counts)."""
        mine = _slot_view(records)[first:last, :, :4].contiguous()
        if not self.active:
            return mine.cpu().numpy()
        if self.stage_through_host and mine.is_cuda:
            mine = mine.cpu()
        rows = [torch.empty_like(mine) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(rows, mine, group=self.group)
        return torch.cat(rows, dim=1).cpu().numpy()

    def reduce_records(self, records, first, last, energies=True):
        """all-reduce records [first, last) slot by slot: the packed maxima (non-negative as int64) with MAX
        (idempotent), and -- exactly once per record -- the three energies (float64 bit patterns) with SUM"""
        if not self.active or last <= first:
            return
        stage = self.stage_through_host and records.is_cuda
        slots = _slot_view(records)
        mx = slots[first:last, :, 0].contiguous()
        if stage:
            mx = mx.cpu()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        slots[first:last, :, 0] = mx.to(records.device)
        if energies:
            en = slots[first:last, :, 1:4].contiguous().view(torch.float64)
            if stage:
                en = en.cpu()
            dist.all_reduce(en, op=dist.ReduceOp.SUM, group=self.group)
            slots[first:last, :, 1:4] = en.view(torch.int64).to(records.device)
