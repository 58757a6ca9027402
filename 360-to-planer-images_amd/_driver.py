"""Device-side scheduling for the drop-in tool: how images and views are dealt to GPUs and how PCIe copies hide
behind kernels.  No pixel arithmetic here -- that is libp2p_hip.so; no collective either: every (image, yaw,
pitch) view is independent work, exactly as the reference treats its per-yaw tasks
(/root/reference/app/panorama_to_plane-pitch.py:252-265) and its per-image loop (P:330-341).

  DevicePipeline   one GPU, two resident jobs used alternately: while image k is resampled, image k+1 is already
                   uploading (its own stream) and the views of image k-1 are downloading (a third stream)
                   -- the cv2.imread (P:244) / cv2.imwrite (P:277) boundary of the reference, overlapped;
  shard_views      the (yaw x pitch) view list of ONE image cut into N contiguous runs, pitch-major, so that a
                   device's share falls into few (pitch, yaw subset) groups and keeps few pitch plans;
  process_views_sharded
                   one image on several GPUs: one host thread per device, ONE job per device that draws exactly the
                   device's (yaw, pitch) views (a view mask: p2p_job_set_view_mask), results stitched on the host.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

try:
    from . import _native
except ImportError:  # the tool file executed as a script (python 360-to-planer-images_amd/panorama_to_plane_pitch.py ...): its siblings are plain modules
    import _native  # type: ignore


def shard_round_robin(n_items, world, rank):
    """Items (panoramas, or views when there are fewer panoramas than GPUs) dealt round-robin."""
    return list(range(rank, n_items, world))


def shard_blocks(n_items, world, rank):
    """Items 0..n-1 cut into `world` contiguous runs whose lengths differ by at most one (the longer ones first)."""
    base, rem = divmod(int(n_items), int(world))
    first = rank * base + min(rank, rem)
    return list(range(first, first + base + (1 if rank < rem else 0)))


# A rank's job sets every tile of a pitch view up once for all the rank's yaws of that view: in units of one view's
# drawing, 2.4 (csrc: a workgroup's set-up against one (panorama, yaw) pair; measured on config 2: 8 ranks, tools/sharded_rank_times.py).  A view's own weight
# grows with its footprint in the panorama, 1 / sin(pitch).
_SETUP_PER_PITCH_GROUP = 2.4


def _run_cost(first, last, n_yaw, weights):
    """Cost of the views first .. last - 1 of the pitch-major list as ONE rank's job."""
    cost, v = 0.0, first
    while v < last:
        p = v // n_yaw
        end = min(last, (p + 1) * n_yaw)
        cost += _SETUP_PER_PITCH_GROUP + (end - v) * weights[p]
        v = end
    return cost


def shard_cost_runs(n_yaw, n_pitch, world, pitch_deg=None):
    """The pitch-major view list cut into `world` contiguous runs [(first, last)] that minimise the costliest run
    (_run_cost): a run that crosses from one pitch view to the next pays two set-ups and gets fewer views for it.  36
    views on 8 ranks by count are 5 5 5 5 4 4 4 4 -- and the third run, two views of one pitch and three of the next,
    is the slowest by a quarter; by cost they are 5 5 3 5 5 4 5 4 ... (tools/sharded_rank_times.py)."""
    import math

    n = n_yaw * n_pitch
    w = [1.0] * n_pitch
    if pitch_deg is not None:
        w = [1.0 / max(0.2, math.sin(math.radians(float(p)))) for p in pitch_deg]

    def greedy(bound):
        runs, a = [], 0
        while a < n:
            b = a + 1
            while b < n and _run_cost(a, b + 1, n_yaw, w) <= bound:
                b += 1
            runs.append((a, b))
            a = b
        return runs

    # the optimum is the cost of some run: try them in ascending order (n <= a few hundred views)
    bounds = sorted({_run_cost(a, b, n_yaw, w) for a in range(n) for b in range(a + 1, min(n, a + 2 * (n // world + n_yaw)) + 1)})
    lo, hi = 0, len(bounds) - 1
    while lo < hi:
        mid = (lo + hi) // 2
        if len(greedy(bounds[mid] + 1e-9)) <= world:
            hi = mid
        else:
            lo = mid + 1
    runs = greedy(bounds[lo] + 1e-9)
    return runs + [(n, n)] * (world - len(runs))


def shard_views(n_yaw, n_pitch, world, rank, how="auto", pitch_deg=None):
    """This rank's views of one image as {pitch index: [yaw indices]}.  The pitch-major list (p0,y0), (p0,y1) ...
    (p1,y0) ... is cut into contiguous runs, one per rank ("blocks"): consecutive yaws of ONE pitch view (of two where
    a run crosses a pitch boundary), so that the rank's masked job sets every tile of that pitch view up once for all
    its yaws.  Config 2's 36 views, slowest rank's launch: 3 ranks 38.0 us (dealt round-robin: 40.0), 6 ranks 23.4
    (28.5), 8 ranks 23.6 (28.0), 12 ranks 17.0 (23.7) -- tools/sharded_rank_times.py.  With fewer ranks than pitch
    views and a yaw count they divide ("auto" then deals round-robin) every rank gets the same yaws of EVERY pitch
    view, a full grid without a mask: 2 ranks 50.4 us against 54.6 in blocks."""
    if how == "auto":
        how = "round_robin" if (world < n_pitch and n_yaw % world == 0) else ("cost" if n_yaw * n_pitch <= 4096 else "blocks")
    if how == "cost":
        # ("blocks" cuts by view COUNT; "cost" by what a run costs its rank: shard_cost_runs)
        first, last = shard_cost_runs(n_yaw, n_pitch, world, pitch_deg)[rank]
        mine = range(first, last)
    else:
        mine = (shard_round_robin if how == "round_robin" else shard_blocks)(n_yaw * n_pitch, world, rank)
    groups = {}
    for v in mine:
        groups.setdefault(v // n_yaw, []).append(v % n_yaw)
    return groups


TILE_ROWS = 16  # rows of an output tile (csrc/p2p_device.h: TILE_H): the unit p2p_job_set_rows deals in


def shard_rows(oh, world, pitch_deg=None, fov_deg=90.0, ow=None):
    """One image's output ROWS cut into `world` contiguous bands [(row0, row1)] of whole tile rows -- every rank draws
    ALL the views, a band of each (p2p_job_set_rows).  A tile's set-up is then spread over all the (panorama, yaw) pairs
    again (a rank's share of whole views has a few yaws per tile) and the number of views no longer caps the speed-up.
    A tile row's weight: 1 / sin(polar angle of the row's centre) summed over the pitch views (the footprint in the
    panorama grows towards a pole); the bands are cut where the running weight passes k / world of the total.  Ranks
    beyond the number of tile rows get (0, 0): nothing."""
    import math

    oh, world = int(oh), int(world)
    n = (oh + TILE_ROWS - 1) // TILE_ROWS
    pitches = [90.0] if not pitch_deg else [float(p) for p in pitch_deg]
    focal = (float(ow if ow else oh) / 2.0) / math.tan(math.radians(min(max(float(fov_deg), 1.0), 179.0)) / 2.0)
    w = []
    for r in range(n):
        yc = min(oh, r * TILE_ROWS + TILE_ROWS / 2.0) - oh / 2.0
        off = math.degrees(math.atan2(yc, focal))
        w.append(sum(1.0 / max(0.2, math.sin(math.radians(min(max(p + off, 1.0), 179.0)))) for p in pitches))
    total, cuts, acc, k = sum(w), [0], 0.0, 1
    for r in range(n):
        acc += w[r]
        while k < world and acc >= total * k / world and len(cuts) < world:
            # (the cut goes behind row r unless stopping before it is closer to the target)
            cuts.append(r + 1 if (acc - total * k / world) <= w[r] / 2.0 or cuts[-1] == r else r)
            k += 1
    while len(cuts) < world:
        cuts.append(n)
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(min(oh, a * TILE_ROWS), min(oh, b * TILE_ROWS)) for a, b in zip(cuts[:-1], cuts[1:])]


def rank_view_set(n_yaw, n_pitch, world, rank, how="auto", pitch_deg=None):
    """The ONE masked job a rank draws its share of an image with: (yaw_idx, pitch_idx, mask, mine).  yaw_idx and
    pitch_idx are the image's yaw / pitch indices that occur in the rank's share (ascending: the job's own angle
    lists), mask is uint8 [len(yaw_idx)][len(pitch_idx)] with 1 where the combination is the rank's own (the layout
    p2p_job_set_view_mask takes), mine the share as (yaw, pitch) image indices in pitch-major order.  A rank with no
    views gets ([], [], empty mask, [])."""
    groups = shard_views(n_yaw, n_pitch, world, rank, how, pitch_deg)
    yaw_idx = sorted({y for ys in groups.values() for y in ys})
    pitch_idx = sorted(groups)
    mine = [(y, p) for p in pitch_idx for y in groups[p]]
    mask = np.zeros((len(yaw_idx), len(pitch_idx)), np.uint8)
    for y, p in mine:
        mask[yaw_idx.index(y), pitch_idx.index(p)] = 1
    return yaw_idx, pitch_idx, mask, mine


class _Ticket:
    """One image in flight on a DevicePipeline; result() waits for its download."""

    def __init__(self, job, views):
        self._job, self._views = job, views

    def result(self):
        if self._job is not None:
            self._job.wait()
            self._job = None
        return self._views


class DevicePipeline:
    """Two resident jobs on one device, used alternately, copies asynchronous.  Not thread-safe: one submitting
    thread per pipeline (the tool keeps one per device)."""

    def __init__(self, device=0, slots=2):
        self.device = int(device)
        self.ctx = _native.Context(self.device)
        self.slots = [None] * int(slots)  # (key, Job, last ticket)
        self.turn = 0

    def submit(self, pano, yaws, pitches, fov, ow, oh, flags=0, maps=None):
        """Enqueue upload -> kernel -> download of every (yaw, pitch) view of `pano`; returns a ticket at once.
        maps = (U, V, maps_key): the tool's exact mode -- the job draws from these pitch maps ([n_pitch][oh][ow] float32,
        the reference's get_pitch_mapping outputs, P:55-73) instead of evaluating its own; they are uploaded when a slot's
        job is created and stay with it, as the reference's pitch_mapping_cache keeps them from image to image (P:18)."""
        pano = _native.as_image(pano, "pano_image")
        ph, pw = pano.shape[:2]
        key = (pw, ph, tuple(float(y) for y in yaws), tuple(float(p) for p in pitches), float(fov), int(ow), int(oh), int(flags),
               None if maps is None else int(maps[2]))
        i = self.turn % len(self.slots)
        self.turn += 1
        slot = self.slots[i]
        if slot is not None:
            slot[2].result()  # the slot's previous image must be out before its buffers are reused
            if slot[0] != key:
                slot[1].close()
                slot = None
        if slot is None:
            job = _native.Job(self.ctx, pw, ph, 1, key[2], key[3], fov, ow, oh, flags=flags)
            if maps is not None:
                try:
                    job.set_maps(None, maps[0], maps[1])
                except Exception:
                    job.close()
                    self.slots[i] = None
                    raise
        else:
            job = slot[1]
        job.set_pano(0, pano, wait=False)
        job.run()
        views = job.get_views_async(0)
        ticket = _Ticket(job, views)
        self.slots[i] = (key, job, ticket)
        return ticket

    def close(self):
        for slot in self.slots:
            if slot is not None:
                try:
                    slot[2].result()
                finally:
                    slot[1].close()
        self.slots = [None] * len(self.slots)
        self.ctx.close()


# A device slot = (device, k): the k-th use of that device inside one call's `devices` list (a device may be named twice:
# two contexts on one GPU).  Slots do NOT depend on the rank a device has in a call: a caller whose device list changes
# order or length from image to image keeps at most (devices x multiplicity) contexts alive, not one per (rank, device)
# pair it has ever used (round 4's keying: 64 possible slots on 8 GPUs, each a HIP stream, cached plans and one of the
# P2P_MAX_CONTEXTS).
_ctx_lock = threading.Lock()  # guards the three dicts below
_ctxs = {}
_groups = {}  # device slot -> (geometry key incl. rank and world, [(job, page-locked staging views, None)]): the slot's resident job
_slot_locks = {}  # device slot -> lock held for the whole of one image on that slot: two concurrent calls that name the
                  # same device take turns instead of interleaving set_pano / run / get_views on the same jobs


def device_slots(devices):
    """[(device, k)] for a call's device list: k counts the earlier occurrences of the same device."""
    seen, out = {}, []
    for d in devices:
        d = int(d)
        out.append((d, seen.get(d, 0)))
        seen[d] = seen.get(d, 0) + 1
    return out


def live_sharded_contexts():
    """How many contexts the sharded path keeps right now (tests: bounded by the devices in use, whatever their order)."""
    with _ctx_lock:
        return len(_ctxs)


def _slot_lock(slot):
    with _ctx_lock:
        lk = _slot_locks.get(slot)
        if lk is None:
            lk = _slot_locks[slot] = threading.Lock()
        return lk


def _close_group(slot):
    """Close the jobs a slot keeps (borrowers before the owner).  The caller holds the slot's lock."""
    with _ctx_lock:
        kept = _groups.pop(slot, None)
    if kept is not None:
        for job, _, _ in reversed(kept[1]):
            job.close()


def _shared_ctx(slot):
    """One context per (device slot) for the sharded path, kept for the life of the process -- and with it the
    plans and yaw tables of every geometry it has seen (the context's table caches)."""
    with _ctx_lock:
        c = _ctxs.get(slot)
        if c is None:
            c = _ctxs[slot] = _native.Context(slot[0])
        return c


def release_sharded():
    """Free the jobs and contexts process_views_sharded keeps between images."""
    with _ctx_lock:
        slots = list(_groups) + [s for s in _ctxs if s not in _groups]
    for slot in slots:
        with _slot_lock(slot):
            _close_group(slot)
            with _ctx_lock:
                c = _ctxs.pop(slot, None)
            if c is not None:
                c.close()


def process_views_sharded(pano, yaws, pitches, ow, oh, fov, devices, flags=0, how="auto", maps=None):
    """Every (yaw, pitch) view of ONE panorama drawn by several GPUs; each device uploads the panorama once, draws its
    share and downloads it; the host stitches [n_yaw][n_pitch][oh][ow][3].  how = "rows": every device draws a band of
    rows of EVERY view (shard_rows, p2p_job_set_rows: a tile's set-up spread over all the yaws, no cap by the number of
    views -- config 2 on 8 ranks: 14 us per rank against 19.8); "views": the pitch-major view list in contiguous runs
    (shard_views: a masked job per device); "auto": rows when there is a tile row per device.
    `devices` may name a device twice (two contexts on one GPU).
    maps = (U, V, maps_key): the tool's exact mode (DevicePipeline.submit) -- every device's job draws from these pitch maps.
    A device slot keeps its jobs (device buffers, plan, yaw tables) for the next image of the same geometry, as the
    reference keeps its maps from image to image (P:17-18): a second image pays uploads, view kernels and downloads."""
    pano = _native.as_image(pano, "pano_image")
    ph, pw = pano.shape[:2]
    yaws, pitches = [float(y) for y in yaws], [float(p) for p in pitches]
    out = np.empty((len(yaws), len(pitches), int(oh), int(ow), 3), dtype=np.uint8)
    world = len(devices)
    slots = device_slots(devices)
    if how == "auto":
        how = "rows" if world > 1 and (int(oh) + TILE_ROWS - 1) // TILE_ROWS >= world else "views"
    bands = shard_rows(oh, world, pitches, fov, ow) if how == "rows" else None
    maps_id = None if maps is None else int(maps[2])

    # slots of earlier calls that this call does not use (other devices, a second context on one device): their jobs
    # hold panoramas and views, their contexts a stream and cached plans -- both go, unless another call is inside the
    # slot right now (its lock is taken: that call's own sweep, or release_sharded(), gets it later)
    with _ctx_lock:
        stale = [s for s in set(_groups) | set(_ctxs) if s not in slots]
    for slot in stale:
        lk = _slot_lock(slot)
        if not lk.acquire(blocking=False):
            continue
        try:
            _close_group(slot)
            with _ctx_lock:
                c = _ctxs.pop(slot, None)
            if c is not None:
                c.close()
        finally:
            lk.release()

    def one_device(rank):
        slot = slots[rank]
        with _slot_lock(slot):
            _one_device_locked(rank, slot)

    def _rows_locked(rank, slot):
        # this device's band of rows of every view: ONE job over all yaws and pitches, planned for the band alone
        r0, r1 = bands[rank]
        geo = (pw, ph, tuple(yaws), tuple(pitches), float(fov), int(ow), int(oh), int(flags), world, rank, "rows", r0, r1, maps_id)
        with _ctx_lock:
            kept = _groups.get(slot)
        if kept is not None and kept[0] != geo:
            _close_group(slot)
            kept = None
        if r1 <= r0:
            return
        ctx = _shared_ctx(slot)
        n_views = len(yaws) * len(pitches)
        if kept is None:
            job = _native.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh, flags=flags)
            try:
                job.set_rows(r0, r1)
                if maps is not None:
                    job.set_maps(None, maps[0], maps[1])
                try:
                    stage = _native.pinned_empty((n_views, r1 - r0, int(ow), 3))
                except (MemoryError, _native.P2PError, OSError):
                    stage = np.empty((n_views, r1 - r0, int(ow), 3), np.uint8)
            except Exception:
                job.close()
                raise
            with _ctx_lock:
                _groups[slot] = (geo, [(job, stage, None)])
        with _ctx_lock:
            job, stage, _ = _groups[slot][1][0]
        try:
            job.set_pano(0, pano, wait=False)  # once per device
            job.run()
            packed = job.ow % 4 == 0  # (odd widths go through the job's packing buffer: one view at a time)
            for y in range(len(yaws)):
                for p in range(len(pitches)):
                    k = y * len(pitches) + p
                    if packed:
                        job.get_view_rows_async(y, p, r0, r1, stage[k])
                    else:
                        stage[k] = job.get_view_rows(y, p, r0, r1)
            job.wait()
            out[:, :, r0:r1] = stage.reshape(len(yaws), len(pitches), r1 - r0, int(ow), 3)
        except Exception:
            _close_group(slot)
            raise

    def _one_device_locked(rank, slot):
        if bands is not None:
            return _rows_locked(rank, slot)
        geo = (pw, ph, tuple(yaws), tuple(pitches), float(fov), int(ow), int(oh), int(flags), world, rank, maps_id)
        yaw_idx, pitch_idx, mask, mine = rank_view_set(len(yaws), len(pitches), world, rank, pitch_deg=pitches)
        with _ctx_lock:
            kept = _groups.get(slot)
        if kept is not None and kept[0] != geo:
            _close_group(slot)  # (also when this geometry leaves the rank with nothing to draw)
            kept = None
        if not mine:
            return
        ctx = _shared_ctx(slot)
        # ONE job per device: the yaws and pitches that occur in the rank's share, and a view mask for the combinations
        # that are really its own (36 views on 8 ranks: 4 or 5 of a 3 x 3 grid) -- one launch, whose pitch views share
        # the source rows they read, instead of one job and one launch per pitch
        if kept is None:
            job = _native.Job(ctx, pw, ph, 1, [yaws[y] for y in yaw_idx], [pitches[p] for p in pitch_idx], fov, ow, oh, flags=flags)
            try:
                if not mask.all():
                    job.set_view_mask(mask)
                if maps is not None:  # (the job's pitch list is the rank's subset of the image's)
                    job.set_maps(None, np.ascontiguousarray(maps[0][pitch_idx]), np.ascontiguousarray(maps[1][pitch_idx]))
                try:
                    stage = _native.pinned_empty((len(mine), int(oh), int(ow), 3))
                except (MemoryError, _native.P2PError, OSError):
                    stage = np.empty((len(mine), int(oh), int(ow), 3), np.uint8)
            except Exception:
                job.close()
                raise
            with _ctx_lock:
                _groups[slot] = (geo, [(job, stage, None)])
        with _ctx_lock:
            job, stage, _ = _groups[slot][1][0]
        try:
            job.set_pano(0, pano, wait=False)  # once per device
            job.run()
            packed = job.ow % 4 == 0  # (odd widths go through the job's packing buffer: one view at a time)
            for k, (y, p) in enumerate(mine):
                if packed:
                    job.get_view_async(yaw_idx.index(y), pitch_idx.index(p), stage[k])
                else:
                    stage[k] = job.get_view(yaw_idx.index(y), pitch_idx.index(p))
            job.wait()
            for k, (y, p) in enumerate(mine):
                out[y, p] = stage[k]
        except Exception:
            _close_group(slot)
            raise

    with ThreadPoolExecutor(max_workers=world) as ex:
        for f in [ex.submit(one_device, r) for r in range(world)]:
            f.result()
    return out
