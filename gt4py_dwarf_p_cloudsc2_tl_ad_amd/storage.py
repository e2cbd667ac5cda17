"""Field storages of the MI355X build: PyTorch-ROCm tensors, physical layout [level][column].

The reference's fields are GT4Py storages of logical shape ``(nx, ny, nz+1)`` whose ``.data`` the
harnesses index as ``[:, 0, :]`` (/root/reference/src/cloudsc2_gt4py/physics/tangent_linear/validation.py:243-249,
adjoint/validation.py:217-220).  Here a field is ONE contiguous ``(nz+1, nx)`` allocation in HBM
(column index fastest, so a wave64 reading one level of 64 adjacent columns issues one coalesced
request) exposed as the permuted view ``(nx, 1, nz+1)`` - the same indexing works unchanged.

All 3-D storages have nz+1 levels, as in the reference (every kernel runs on
``domain=(nx, 1, nz+1)``, nonlinear/microphysics.py:168-169); K-vectors (`f_eta`, `klevel`) have nz+1
entries too.

Placement in HBM (`FieldArena`, `tune_placement`; docs/TUNING_LOG.md 3.7).  A stencil call streams 26 (NL) to 72 (AD) fields
concurrently, every wave touching the same (level, column) offset of each of them at about the same time, so how the
fields' starting addresses relate decides how those requests fall onto HBM channels, banks and rows.  Measured with the
kernels unchanged (profiles/r02/placement_*.txt, layout_scan*.txt): the same fields run cloudsc2_nl anywhere between 295
and 350 us depending on placement alone; separate `torch` allocations (the default: `zeros` / `from_klayout`) land anywhere
in that range from process to process; slabs of ONE allocation behave the same in every process, and a per-field stagger
of 2 304 B is worth 3 % over none, but which spacing between the slabs is fastest depends on where the driver put the
arena - so `tune_placement` measures it, with the caller's own kernel sequence as the objective (bench.py does, before its
timed region).  `CLOUDSC2_FIELD_ARENA=<n>` makes `zeros` / `from_klayout` draw GPU fields from automatic arenas of n slabs
(off by default: the default arena layout is reproducible, not faster on average than separate allocations).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

_TORCH_DTYPES = {
    np.dtype("float64"): torch.float64,
    np.dtype("float32"): torch.float32,
    np.dtype("int64"): torch.int64,
    np.dtype("int32"): torch.int32,
}


def torch_dtype(dtype: Any) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    return _TORCH_DTYPES[np.dtype(dtype)]


#: every level of a field starts on a 512-byte boundary of its storage: one wave's request for 64 fp64 columns.  For nx
#: a multiple of 64 (fp64) / 128 (fp32) columns - every BASELINE size - that is the dense layout; for any other nx the
#: level pitch is padded, which keeps rows aligned for the LDS-DMA load path (cloudsc2_nl takes its ring kernel for ANY
#: nx then; measured at 65 500 fp64 columns: 330 us padded against 378-388 us dense and 410 us on the register path)
ROW_ALIGN_BYTES = 512


def level_pitch(nx: int, dtype: Any) -> int:
    """columns from one level to the next (lev_stride) of the storages this module allocates: nx rounded up so that a level
    is a multiple of ROW_ALIGN_BYTES"""
    item = torch.empty((), dtype=torch_dtype(dtype)).element_size()
    per = max(1, ROW_ALIGN_BYTES // item)
    return -(-int(nx) // per) * per


def _padded_kc(nx: int, nz: int, dt: torch.dtype, dev: torch.device) -> torch.Tensor:
    """zero-initialised (nz+1, nx) [level][column] window of a (nz+1, level_pitch) allocation"""
    ls = level_pitch(nx, dt)
    return torch.zeros((nz + 1, ls), dtype=dt, device=dev)[:, :nx]


def logical_view(kc: torch.Tensor) -> torch.Tensor:
    """(nz+1, nx) physical tensor -> (nx, 1, nz+1) logical view (no copy)."""
    if kc.dim() != 2:
        raise ValueError(f"expected a 2-D [level][column] tensor, got shape {tuple(kc.shape)}")
    return kc.unsqueeze(1).permute(2, 1, 0)


def klayout(field: torch.Tensor) -> torch.Tensor:
    """(nx, 1, nz+1) logical view -> (nz+1, nx) physical view (no copy)."""
    if field.dim() != 3 or field.shape[1] != 1:
        raise ValueError(f"expected a (nx, 1, nz+1) field, got shape {tuple(field.shape)}")
    return field.permute(2, 1, 0).squeeze(1)


class FieldArena:
    """`capacity` field slabs of shape (nz+1, nx) in ONE allocation: slab i starts at a 2 MB boundary of the address space
    + (i * stagger) mod 64 KB.  `zeros()` hands out the next slab as a zero-initialised logical (nx, 1, nz+1) field."""

    SLAB_ALIGN = 2 << 20
    STAGGER = 2304            # bytes; 9 x 256: coprime with the 256 channel slots of a 64 KB window
    STAGGER_WRAP = 65536

    def __init__(self, nx: int, nz: int, dtype: Any, device: Any, capacity: int, stagger: Optional[int] = None,
                 extra_spacing: int = 0) -> None:
        self.nx, self.nz, self.capacity = int(nx), int(nz), int(capacity)
        self.dtype = torch_dtype(dtype)
        self.device = torch.device(device)
        self.stagger = self.STAGGER if stagger is None else int(stagger)
        item = torch.empty((), dtype=self.dtype).element_size()
        if self.stagger % 16 or self.capacity < 1:
            raise ValueError("stagger must be a multiple of 16 bytes (the kernels' 16-byte load paths), capacity >= 1")
        self.ls = level_pitch(self.nx, self.dtype)
        fbytes = (self.nz + 1) * self.ls * item
        self.slab = -(-(fbytes + self.STAGGER_WRAP) // self.SLAB_ALIGN) * self.SLAB_ALIGN + int(extra_spacing)
        self._item = item
        self._buf = torch.zeros((self.capacity * self.slab + self.SLAB_ALIGN) // item, dtype=self.dtype, device=self.device)
        self._base = (-self._buf.data_ptr()) % self.SLAB_ALIGN
        self._next = 0

    @property
    def free_slots(self) -> int:
        return self.capacity - self._next

    def offset_of(self, slot: int) -> int:
        """byte offset of slab `slot` from the arena's first 2 MB boundary"""
        return slot * self.slab + (slot * self.stagger) % self.STAGGER_WRAP

    def zeros(self) -> torch.Tensor:
        if self._next >= self.capacity:
            raise RuntimeError(f"FieldArena is full ({self.capacity} fields)")
        o = (self._base + self.offset_of(self._next)) // self._item
        self._next += 1
        n = (self.nz + 1) * self.ls
        return logical_view(self._buf[o:o + n].view(self.nz + 1, self.ls)[:, :self.nx])


def plan_placement_grid(n: int, slab: int, free_bytes: Optional[int], *, spacings, staggers, shifts_mb, wide_spacings,
                        wide_shifts_mb, max_arena_bytes: int, max_shift_spans: float):
    """The candidate placements `tune_placement` will time for `n` fields of `slab` bytes, and the size of the ONE arena
    that holds them all: never more than `max_arena_bytes` and never more than 60 % of `free_bytes` (what the device
    has free NOW - on a node with one process per GPU that is the caller's own device, so eight ranks plan eight
    independent arenas).  Returns (grid of (extra spacing x 2 MB, stagger, shift bytes), arena bytes, spacings kept);
    raises RuntimeError when not even the densest placement fits."""
    two_mb = FieldArena.SLAB_ALIGN
    free_cap = int(0.6 * free_bytes) if free_bytes is not None else max_arena_bytes
    max_arena_bytes = min(max_arena_bytes, free_cap)        # never ask for more than 60 % of what is free now
    fit = [e for e in spacings if n * (slab + e * two_mb) + two_mb <= max_arena_bytes]
    if not fit:
        raise RuntimeError(f"tune_placement: {n} fields of {slab} B do not fit the arena cap of {max_arena_bytes} B "
                           "(min(max_arena_bytes, 60 % of the free device memory))")
    spacings = tuple(fit)
    emax = max(spacings)
    span = n * (slab + emax * two_mb) + two_mb
    shifts = [int(sh) << 20 for sh in shifts_mb
              if span + (int(sh) << 20) <= max_arena_bytes and (int(sh) << 20) <= max_shift_spans * span] or [0]
    grid = [(e, st, sh) for sh in shifts for e in spacings for st in staggers]
    grid += [(e, st, int(sh) << 20) for e in wide_spacings for sh in wide_shifts_mb for st in staggers
             if n * (slab + e * two_mb) + two_mb + (int(sh) << 20) <= max_arena_bytes]
    arena_need = max(n * (slab + e * two_mb) + two_mb + sh for e, _, sh in grid)
    return grid, arena_need, spacings


def _without_timing_brackets(fn):
    """Run `fn` with the drivers' timing brackets switched off: an objective that runs a whole harness (TaylorTest.run opens
    ~23 brackets per run) would otherwise pile up thousands of pending event pairs and have `Timer._resolve()` - a device-wide
    synchronisation - land inside some candidate's timed window."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        from .framework import timing as _timing

        was = _timing.set_enabled(False)
        try:
            return fn(*args, **kwargs)
        finally:
            _timing.set_enabled(was)

    return wrapper


@_without_timing_brackets
def tune_placement(nx: int, nz: int, dtype: Any, device: Any, order, sources, launch, *, spacings=tuple(range(0, 64)),
                   staggers=(FieldArena.STAGGER, 8448), shifts_mb=(0, 4096, 8192, 12288), wide_spacings=(),
                   wide_shifts_mb=tuple(range(0, 32769, 2048)), launches: int = 5, rounds: int = 3,
                   budget_s: float = 4.0, max_arena_bytes: int = 40 << 30, max_shift_spans: float = 4.0,
                   keep_all: bool = False, extend_shifts_mb=(16384, 20480, 24576, 28672), extend_below_gain: float = 0.04,
                   extend_min_span_bytes: int = 1 << 30):
    """Calibrate WHERE the fields of a stencil call sit in HBM, for this process.

    The rate at which a call streams its 26-72 fields depends on how their starting addresses relate (channel, bank and
    row bits of 26+ concurrent streams) - by ~10 % between good and bad relations - and the best spacing between field
    starts differs from process to process, because it depends on where the driver put the arena's physical pages
    (profiles/layout_scan.py, profiles/r02/layout_scan*.txt: +3 x 2 MB was best in one process, +29 x 2 MB in another,
    295-298 us against 317 us for the default spacing).  So it is measured: ONE arena is allocated; for every candidate
    (slab spacing = minimal 2-MB-aligned slab + e x 2 MB, stagger s, and a shift of the whole placement inside the
    arena) the fields named in `order` are placed at shift + i x spacing + (i x s) mod 64 KB, `sources[name]`
    ([level][column] tensors, or None for outputs) are copied in and `launch(fields)` - the caller's real kernel
    sequence on those fields - is timed with HIP events (median of `rounds` x `launches`; `budget_s` caps the GPU time
    spent, so slow sequences and big fields try an evenly spread subset of the candidates).

    Candidates: the full grid `spacings` x `staggers` x `shifts_mb` (whole-placement shifts, as far as the arena may
    grow) and, when `wide_spacings` is given, a second family with 0.25-1 GB between field starts at every shift of
    `wide_shifts_mb`.  The wide family is OFF by default: the speed of a placement has a ~32 GB period in the arena
    offset on some leases (placements that straddle such a boundary with their last few fields beyond it run the NL
    kernel 4-8 % faster, profiles/r02/placement_structure.txt) and a wide placement meets a boundary more often than a
    narrow one - but over fresh processes it did not find faster placements than the narrow grid (same file, A/B).

    Second stage (r03): the fast placements of an arena are the ones whose last fields lie beyond a junction of the arena's
    physical backing, and on some leases the first junction is ~28-32 GB into the arena (profiles/r02/placement_structure.txt,
    scans 1 and 2: fast shifts at 28, 29 | 60, 61 | ... GB) - out of reach of shifts 0-12 GB.  When the first stage gains less
    than `extend_below_gain` over the default placement, a second arena is searched at `extend_shifts_mb` (same budget
    again), the two winners are timed against each other, and the loser's arena is freed.  A lease that needs it pays
    ~4 s and a 33-39 GB arena instead of 18-24 GB; a lease that does not, nothing.

    Returns (fields at the fastest placement - inputs copied in, outputs zeroed; a report dict).  The arena stays alive
    as long as the returned fields do (up to `max_arena_bytes` and never more than 60 % of the free device memory:
    typically 17-23 GB of the 288 GB; while a second stage runs, both arenas are alive - the second is sized against what
    is free AFTER the first was allocated, so the peak is at most 2 x `max_arena_bytes` = 80 GB - and the loser is freed
    before returning).  Like picking a ring depth by grid size, this decides nothing about the
    arithmetic: results are bit-identical for every placement."""
    dt, dev = torch_dtype(dtype), torch.device(device)
    item = torch.empty((), dtype=dt).element_size()
    n = len(order)
    two_mb = FieldArena.SLAB_ALIGN
    ls = level_pitch(nx, dt)
    slab = -(-((nz + 1) * ls * item + FieldArena.STAGGER_WRAP) // two_mb) * two_mb
    # what is free on THIS device now (one process per GPU: every rank sizes its arena against its own device)
    free_bytes = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else None
    grid, arena_need, spacings = plan_placement_grid(
        n, slab, free_bytes, spacings=spacings, staggers=staggers, shifts_mb=shifts_mb, wide_spacings=wide_spacings,
        wide_shifts_mb=wide_shifts_mb, max_arena_bytes=max_arena_bytes, max_shift_spans=max_shift_spans)
    buf = torch.zeros(arena_need // item, dtype=dt, device=dev)
    base = (-buf.data_ptr()) % two_mb
    count = (nz + 1) * ls

    def place(e, st, sh=0):
        fields = {}
        for i, name in enumerate(order):
            o = (base + sh + i * (slab + e * two_mb) + (i * st) % FieldArena.STAGGER_WRAP) // item
            fields[name] = logical_view(buf[o:o + count].view(nz + 1, ls)[:, :nx])
        for name, src in sources.items():
            if src is not None:
                klayout(fields[name]).copy_(src)
        return fields

    def timed(fields):
        for _ in range(2):
            launch(fields)
        ts = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(launches):
                launch(fields)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / launches)
        return sorted(ts)[len(ts) // 2]

    default = (min(spacings), staggers[0])
    # steady state first.  After an idle period the GPU needs 10-15 ms of work to reach its core clocks
    # (profiles/r02/window_probe.txt) and, on this pool, up to a second of sustained streaming before the memory side
    # settles (the same placement measured at the start and at the end of an un-warmed calibration differed by 10 %):
    # candidates timed during either ramp would lose to later ones for that reason alone.  So the default placement is
    # run until two consecutive chunks agree to 1 % (at most ~1.5 s).
    warm = place(*default)

    def chunk_ms(nl_):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(nl_):
            launch(warm)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / nl_

    t1 = chunk_ms(1)
    per_chunk = max(2, min(100, int(60.0 / max(t1, 1e-3))))      # ~60 ms of work per chunk
    prev, spent = chunk_ms(per_chunk), 0.0
    while spent < 1500.0:
        cur = chunk_ms(per_chunk)
        spent += cur * per_chunk
        if abs(cur - prev) <= 0.01 * prev:
            break
        prev = cur
    t_default = timed(warm)
    default = (default[0], default[1], 0)
    results = [(t_default,) + default]
    cands = [c for c in grid if c != default]
    # as many candidates as fit `budget_s` of GPU time (big fields: fewer), spread evenly over the list
    per_cand = (2 + rounds * launches) * t_default * 1e-3 * 1.3
    keep = max(7, min(len(cands), int(budget_s / max(per_cand, 1e-6))))
    if keep < len(cands):
        cands = [cands[round(i * (len(cands) - 1) / (keep - 1))] for i in range(keep)]
    for c in cands:
        results.append((timed(place(*c)),) + c)
    t_best = min(results)[0]
    # second pass over the eight fastest (residual drift and single lucky measurements), then the winner against the
    # default once more
    finals = sorted((timed(place(*c[1:])),) + c[1:] for c in sorted(results)[:8])
    best = finals[0][1:]
    t_best2, t_default2 = timed(place(*best)), timed(place(*default))
    if t_best2 >= t_default2:
        best, t_best2 = default, t_default2
    e_best, st_best, sh_best = best
    fields = place(*best)

    def restore(fs):
        """the documented state of returned fields: inputs equal `sources`, output-only fields zero - whatever ran on them"""
        for name in order:
            src = sources.get(name)
            if src is None:
                fs[name].zero_()
            else:
                klayout(fs[name]).copy_(src)

    restore(fields)
    report = {"candidates": len(results), "default_ms": t_default2, "tuned_ms": t_best2,
              "first_pass_default_ms": t_default, "first_pass_best_ms": t_best,
              "extra_spacing_x2MB": int(e_best), "stagger_bytes": int(st_best), "shift_MB": int(sh_best >> 20),
              "slab_bytes": int(slab), "arena_bytes": int(buf.numel() * item),
              "first_pass_top": [(round(c[0], 4), c[1], c[2], c[3] >> 20) for c in sorted(results)[:8]],
              "second_pass": [(round(c[0], 4), c[1], c[2], c[3] >> 20) for c in finals]}
    if keep_all:                 # every first-pass timing (profiles/placement_distribution.py)
        report["first_pass_all"] = [(round(c[0], 4), c[1], c[2], c[3] >> 20) for c in results]
    # (only for placements that span >= 1 GB: small fields live in the caches, and a 30 GB arena for them would be absurd)
    free_now = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else 0
    room = min(max_arena_bytes, int(0.6 * free_now)) if dev.type == "cuda" else 0
    reachable = [sh for sh in extend_shifts_mb if (int(sh) << 20) + n * slab + two_mb <= room]
    if reachable and n * slab >= extend_min_span_bytes and t_best2 > (1.0 - extend_below_gain) * t_default2:
        try:
            fields2, report2 = tune_placement(
                nx, nz, dtype, device, order, {k: (klayout(fields[k]) if sources.get(k) is not None else None) for k in order},
                launch, spacings=spacings, staggers=staggers, shifts_mb=tuple(reachable), wide_spacings=(),
                launches=launches, rounds=rounds, budget_s=budget_s, max_arena_bytes=max_arena_bytes, max_shift_spans=1e9,
                extend_shifts_mb=())
        except RuntimeError as exc:          # no room for the bigger arena: the first stage stands
            report["second_stage"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            return fields, report
        t1, t2 = timed(fields), timed(fields2)
        restore(fields)          # both winners have just been run 2 + rounds x launches times: back to the contract
        restore(fields2)
        stage = {"tuned_ms": report2["tuned_ms"], "retimed_first_ms": t1, "retimed_second_ms": t2,
                 "shift_MB": report2["shift_MB"], "extra_spacing_x2MB": report2["extra_spacing_x2MB"],
                 "candidates": report2["candidates"], "arena_bytes": report2["arena_bytes"]}
        if t2 < t1:
            report2.update(default_ms=t_default2, first_stage={"tuned_ms": t_best2, "shift_MB": report["shift_MB"],
                                                                "extra_spacing_x2MB": report["extra_spacing_x2MB"]},
                           second_stage=dict(stage, chosen=True), tuned_ms=t2,
                           candidates=report["candidates"] + report2["candidates"])
            del fields, buf, warm, finals
            return fields2, report2
        report["second_stage"] = dict(stage, chosen=False)
        del fields2
    return fields, report


#: automatic arenas behind `zeros` / `from_klayout` for GPU fields: slabs per arena (0 = one torch allocation per field)
_ARENA_CAPACITY = int(os.environ.get("CLOUDSC2_FIELD_ARENA", "0"))
_ARENA_MAX_BYTES = 48 << 30       # an automatic arena never exceeds this; bigger fields get fewer slabs per arena
_arenas: Dict[Tuple[int, int, torch.dtype, torch.device], FieldArena] = {}


def set_arena_capacity(capacity: int) -> int:
    """Slabs per automatic arena (0 disables arenas); returns the previous setting.  Existing fields are unaffected."""
    global _ARENA_CAPACITY
    old, _ARENA_CAPACITY = _ARENA_CAPACITY, int(capacity)
    _arenas.clear()
    return old


def _arena_for(nx: int, nz: int, dtype: torch.dtype, device: torch.device) -> Optional[FieldArena]:
    if _ARENA_CAPACITY <= 0 or device.type != "cuda" or nx <= 0:
        return None
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (nx, nz, dtype, device)
    a = _arenas.get(key)
    if a is None or a.free_slots == 0:
        item = torch.empty((), dtype=dtype).element_size()
        slab = -(-((nz + 1) * level_pitch(nx, dtype) * item + FieldArena.STAGGER_WRAP) // FieldArena.SLAB_ALIGN) * FieldArena.SLAB_ALIGN
        cap = max(1, min(_ARENA_CAPACITY, _ARENA_MAX_BYTES // slab))
        a = _arenas[key] = FieldArena(nx, nz, dtype, device, cap)
    return a


def zeros(nx: int, nz: int, dtype: Any, device: Any) -> torch.Tensor:
    """Zero-initialised 3-D field, logical shape (nx, 1, nz+1); on the GPU a slab of an automatic `FieldArena`."""
    dt, dev = torch_dtype(dtype), torch.device(device)
    arena = _arena_for(int(nx), int(nz), dt, dev) if dt.is_floating_point else None
    if arena is not None:
        return arena.zeros()
    if dev.type == "cuda" and dt.is_floating_point:
        return logical_view(_padded_kc(int(nx), int(nz), dt, dev))
    return logical_view(torch.zeros((nz + 1, nx), dtype=dt, device=dev))


def zeros_k(nz: int, dtype: Any, device: Any) -> torch.Tensor:
    """Zero-initialised K-vector with nz+1 entries."""
    return torch.zeros((nz + 1,), dtype=torch_dtype(dtype), device=device)


def from_klayout(array_kc: Any, dtype: Any, device: Any) -> torch.Tensor:
    """Copy a host/device ``[level][column]`` array into a new field storage."""
    t = torch.as_tensor(array_kc)
    dt, dev = torch_dtype(dtype), torch.device(device)
    if t.dim() == 2 and dt.is_floating_point:
        arena = _arena_for(int(t.shape[1]), int(t.shape[0]) - 1, dt, dev)
        if arena is not None:
            f = arena.zeros()
            klayout(f).copy_(t)
            return f
    if t.dim() == 2 and dt.is_floating_point and dev.type == "cuda":
        kc = _padded_kc(int(t.shape[1]), int(t.shape[0]) - 1, dt, dev)
        kc.copy_(t)
        return logical_view(kc)
    t = t.to(device=dev, dtype=dt).contiguous()
    return logical_view(t)


def field_geometry(field: torch.Tensor) -> Tuple[int, int, int]:
    """(nx, nlev, lev_stride) of a logical (nx, 1, nlev) field; raises if it is not column-fastest."""
    if field.dim() != 3 or field.shape[1] != 1:
        raise ValueError(f"field must have logical shape (nx, 1, nz+1), got {tuple(field.shape)}")
    nx, _, nlev = field.shape
    if nx > 1 and field.stride(0) != 1:
        raise ValueError(
            f"field is not column-fastest (stride over columns = {field.stride(0)}); "
            "allocate it with storage.zeros / storage.from_klayout"
        )
    ls = field.stride(2) if nlev > 1 else nx
    return nx, nlev, ls
