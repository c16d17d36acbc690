"""Replay buffer and small helpers of the learner path (reference: utils.py).

``ReplayBuffer`` keeps the reference's constructor, ``add``, ``sample_cpc``,
``save``/``load``, ``idx``/``full`` -- but the ring lives in HBM (uint8, NHWC so
a cropped row is one contiguous run) and sampling hands the learner *references*
(frame indices + crop offsets) instead of materialised float tensors: the
reference's 3 x B x C x H x W float32 host->device copy per update (utils.py:161-166)
disappears.  Index and crop-offset draws stay on the host in NumPy's legacy
global stream, in the reference's order, so a seeded run samples the same
transitions and windows (bit-exact).

``dedup_frames=True`` stores every RGB frame once: a frame
stack of k frames shares k-1 of them with its successor and ``next_obs[t]`` is
``obs[t+1]`` inside an episode (utils.py:238-268), so an environment step adds
ONE new frame instead of 2k -- 63.5 KB -> 21 KB per transition at 84x84x9,
339 KB -> 85 KB at 168x168x12.  Sharing is detected from the bytes handed to
``add`` (hash, then full comparison), never assumed, so sampled pixels are the
reference's whatever the caller does.
"""
import collections
import os
import random

import numpy as np
import torch

from . import augmentations, ops

try:  # 64-bit frame fingerprints for the de-duplicating store (~10 GB/s); zlib is the slower stand-in
    from xxhash import xxh3_64_intdigest as _fingerprint
except ImportError:  # pragma: no cover
    import zlib

    def _fingerprint(buf):
        return zlib.crc32(buf) | (zlib.adler32(buf) << 32)


def _lib_tracing():
    from . import _lib
    return _lib._trace_hook is not None


class eval_mode(object):
    """Context manager that puts the given models (anything with ``.training`` and ``.train(bool)``) in
    evaluation mode and restores each one's previous mode on exit (utils.py:21-34)."""

    def __init__(self, *models):
        self.models = models
        self._was_training = None

    def __enter__(self):
        self._was_training = [m.training for m in self.models]
        for m in self.models:
            m.train(False)

    def __exit__(self, *exc):
        for m, mode in zip(self.models, self._was_training):
            m.train(mode)
        return False


def soft_update_params(net, target_net, tau):
    """utils.py:37-41, one fused lerp kernel per tensor.  CurlSacAgent uses the
    flat-buffer form (``soft_update_targets``): two launches for all 24 tensors."""
    for param, target_param in zip(net.parameters(), target_net.parameters()):
        ops.soft_update(param.data.view(-1), target_param.data.view(-1), tau)


def set_seed_everywhere(seed):
    """utils.py:44-49."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def make_dir(dir_path):
    """Create ``dir_path`` unless it exists; like the reference (utils.py:61-66) a failure is reported, not raised,
    and the path is returned either way."""
    if not os.path.isdir(dir_path):
        try:
            os.mkdir(dir_path)
        except OSError as e:
            print('Unable to create directory %s (%s)' % (dir_path, e.strerror))
    return dir_path


def module_hash(module):
    """Sum of the sums of a module's state_dict tensors -- the reference's cheap "did the weights change" number
    (utils.py:50-54)."""
    return sum(t.sum().item() for t in module.state_dict().values())


def preprocess_obs(obs, bits=5):
    """Bit-depth reduction with uniform dequantisation noise, arXiv:1807.03039 (utils.py:67-77; unused by the
    learner path, kept so that ``utils.`` resolves every name the reference module has)."""
    assert obs.dtype == torch.float32
    bins = 2 ** bits
    if bits < 8:
        obs = torch.floor(obs / 2 ** (8 - bits))
    return obs / bins + torch.rand_like(obs) / bins - 0.5


class FrameStack(object):
    """Observation wrapper that returns the last ``k`` frames concatenated on the channel axis (utils.py:238-268):
    ``reset`` fills the stack with k copies of the first frame, ``step`` pushes the new frame.  Plain duck-typed
    wrapper (``gymnasium`` is only used for the observation space when it is importable): every other attribute
    is forwarded to the wrapped environment, as ``gym.Wrapper`` does."""

    def __init__(self, env, k):
        self.env = env
        self._k = k
        self._frames = collections.deque([], maxlen=k)
        space = getattr(env, "observation_space", None)
        shp = tuple(getattr(space, "shape", ()))
        if shp:
            stacked = (shp[0] * k,) + shp[1:]
            try:
                import gymnasium
                self.observation_space = gymnasium.spaces.Box(low=0, high=1, shape=stacked, dtype=space.dtype)
            except ImportError:
                self.observation_space = type("Box", (), dict(shape=stacked, dtype=getattr(space, "dtype", np.uint8),
                                                              low=0, high=1))()
        self._max_episode_steps = getattr(env, "_max_episode_steps", None)
        self.curl_driving = False

    def __getattr__(self, name):  # only called for attributes not found on the wrapper itself
        if name in ("env", "_frames"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self):
        first = self.env.reset()
        self.curl_driving = getattr(self.env, "curl_driving", False)
        self._frames.extend([first] * self._k)
        return self._get_obs()

    def step(self, action):
        frame, reward, done, info = self.env.step(action)
        self.env.curl_driving = self.curl_driving
        self._frames.append(frame)
        return self._get_obs(), reward, done, info

    def _get_obs(self):
        assert len(self._frames) == self._k
        return np.concatenate(list(self._frames), axis=0)


class _FrameStore:
    """Host-side bookkeeping of the de-duplicating frame store: reference counts, a free list and a small cache of
    the most recently interned frames (hash -> (frame id, bytes)) that new frames are matched against."""

    def __init__(self, n_frames, recent):
        self.refs = np.zeros(n_frames, dtype=np.int32)
        self.free = collections.deque(range(n_frames))
        self.recent = collections.OrderedDict()  # fingerprint -> (fid, ndarray copy)
        self.max_recent = recent

    def lookup(self, frame):
        """(fid or None, fingerprint) of a (3, H, W) uint8 frame among the recent ones -- byte-exact."""
        h = _fingerprint(frame)
        hit = self.recent.get(h)
        if hit is not None and np.array_equal(hit[1], frame):
            self.recent.move_to_end(h)
            return hit[0], h
        return None, h

    def allocate(self, frame, h):
        if not self.free:
            raise MemoryError("frame store exhausted: the observations handed to add() share fewer frames than a "
                              "frame-stacked episode does; raise frame_capacity or construct the ReplayBuffer with "
                              "dedup_frames=False")
        fid = self.free.popleft()
        self.recent[h] = (fid, np.array(frame, copy=True))
        while len(self.recent) > self.max_recent:
            self.recent.popitem(last=False)
        return fid

    def release(self, fid):
        self.refs[fid] -= 1
        if self.refs[fid] == 0:
            self.free.append(fid)
            for h, (f, _) in list(self.recent.items()):
                if f == fid:
                    del self.recent[h]


class ReplayBuffer(object):
    """Buffer to store environment transitions (utils.py:80-236), HBM-resident."""

    N_SAMPLE_SLOTS = 2  # minibatches whose references may be alive at once (the current one + one drawn ahead)
    EVENT_EVERY = 8     # index uploads per recorded event (16 pinned slots)

    def __init__(self, obs_shape, action_shape, capacity, batch_size, device, augmentor, transform=None,
                 dedup_frames=False, frame_capacity=None):
        self.capacity = capacity
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.augmentor = augmentor
        self.transform = transform
        if len(obs_shape) != 3:
            raise NotImplementedError("curla_amd.ReplayBuffer stores pixel observations (C, H, W) only")
        c, h, w = obs_shape
        self.obs_shape = tuple(obs_shape)
        frame = c * h * w
        A = int(np.prod(action_shape))
        self.dedup_frames = bool(dedup_frames)
        if self.dedup_frames:
            if c % 3 != 0:
                raise ValueError("dedup_frames needs stacked RGB frames (channels a multiple of 3)")
            self._k = c // 3
            if frame_capacity is None:  # one new frame per step + one extra per episode start, with headroom
                frame_capacity = capacity + capacity // 16 + 4 * self._k + 8
            self.frame_capacity = int(frame_capacity)
            total_bytes = self.frame_capacity * 3 * h * w + capacity * (8 * self._k + 4 * A + 8) \
                + self.N_SAMPLE_SLOTS * 2 * batch_size * frame
        else:
            total_bytes = 2 * capacity * frame + capacity * (4 * A + 8)
        if self.device.type == "cuda":
            free, _ = torch.cuda.mem_get_info(self.device)
            if total_bytes > free:
                raise ValueError('Replay buffer size exceeds available memory')  # utils.py:112-113
        dev = self.device
        if self.dedup_frames:
            # every RGB frame once: uint8 [F][H][W][3] (+32 B slack like a ring); a transition keeps 2k frame ids
            f3 = 3 * h * w
            self._frame_store = torch.zeros(self.frame_capacity * f3 + 32, dtype=torch.uint8, device=dev)
            self.frames = self._frame_store[:self.frame_capacity * f3].view(self.frame_capacity, h, w, 3)
            self._fid = torch.zeros((capacity, 2, self._k), dtype=torch.int32, device=dev)
            self._fid_h = np.full((capacity, 2, self._k), -1, dtype=np.int32)
            self._store = _FrameStore(self.frame_capacity, recent=2 * self._k + 2)
            self.obses = self.next_obses = None  # stacks are assembled per minibatch (stack(i) materialises one)
        else:
            # ring storage: NHWC uint8 frames (+32 B slack: the aligning loader reads whole 16-byte runs plus a dword).
            # Both rings live in ONE allocation, next_obs behind obs, so that slot i of next_obs is also frame
            # capacity + i of a single [2 * capacity] ring: a minibatch's obs and next_obs can then be read by one
            # first-layer launch (ObsRef.pair) -- both go through the same online conv weights in the critic phase.
            # (only when the second half then starts on a dword, which the first-layer loader needs of a ring base)
            self._both = None
            if (capacity * frame) % 4 == 0:
                self._ring_store = torch.zeros(2 * capacity * frame + 32, dtype=torch.uint8, device=dev)
                self._obs_store = self._ring_store[:capacity * frame]
                self._next_store = self._ring_store[capacity * frame:2 * capacity * frame]
                self._both = self._ring_store[:2 * capacity * frame].view(2 * capacity, h, w, c)
                self.obses = self._both[:capacity]
                self.next_obses = self._both[capacity:]
            else:
                self._obs_store = torch.zeros(capacity * frame + 32, dtype=torch.uint8, device=dev)
                self._next_store = torch.zeros(capacity * frame + 32, dtype=torch.uint8, device=dev)
                self.obses = self._obs_store[:capacity * frame].view(capacity, h, w, c)
                self.next_obses = self._next_store[:capacity * frame].view(capacity, h, w, c)
        # action | reward | not_done of a transition sit in one row, so add() writes them with one small copy;
        # the three reference attributes are column views of it
        self._n_act = A
        self._sc = torch.empty((capacity, A + 2), dtype=torch.float32, device=dev)
        self.actions = self._sc[:, :A].unflatten(1, tuple(action_shape)) if len(action_shape) != 1 else self._sc[:, :A]
        self.rewards = self._sc[:, A:A + 1]
        self.not_dones = self._sc[:, A + 1:A + 2]
        self.idx = 0
        self.last_save = 0
        self.full = False
        # staging: pinned host rows for add(), device index buffers for sampling
        pin = self.device.type == "cuda"
        # add(): one pinned block per slot = [frames | pad | frame ids, action, reward, not_done], a few slots
        # guarded by events so that add() never waits for the GPU; one device block receives the copy
        self._frame = frame
        n_stage = 2 * frame  # plain: the two stacks; dedup: up to 2k new RGB frames = the same bytes
        self._sc_off = (n_stage + 15) & ~15
        self._hdr = 4 * (2 * self._k) if self.dedup_frames else 0  # frame-id row in front of the scalars
        blk = self._sc_off + self._hdr + 4 * (A + 2)
        self._n_add, self._add_slot = 4, 0
        self._h_add = torch.empty((self._n_add, blk), dtype=torch.uint8, pin_memory=pin)
        self._h_add_np = self._h_add.numpy()
        self._add_events = [None] * self._n_add
        self._d_add = torch.empty(blk, dtype=torch.uint8, device=dev)
        self._d_add_frames = self._d_add[:2 * frame].view(2, frame)
        self._d_add_sc = self._d_add[self._sc_off + self._hdr:].view(torch.float32)
        B = batch_size
        # the host may run several updates ahead of the GPU: a small ring of pinned slots, each guarded by an
        # event, keeps an index upload's source intact until its async copy has executed
        # (an event record is a packet of its own in the stream, ~5 us: one per EVENT_EVERY uses, and a slot is
        # re-written only after the first event recorded at or after its last use has completed)
        self._n_slots, self._slot_use = 16, 0
        nbytes = 2 * B * 8 + B * 4 * 6  # frame indices (obs | next_obs) + the six crop-offset rows
        self._h_index = torch.empty((self._n_slots, nbytes), dtype=torch.uint8, pin_memory=pin)
        self._slot_events = {}
        # pinned slots are read by the GPU in place (ops.sample_stage): no copy-engine transfer in front of an update
        self._h_index_dev = ([ops.host_device_pointer(self._h_index[k]) for k in range(self._n_slots)]
                             if pin and os.environ.get("CURLA_STAGE_COPY", "0") != "1" else None)
        self._staged = False
        # every minibatch gets its own device index block (and, de-duplicated, its own assembled stacks), so the
        # references of one sample stay valid while the next one is drawn (N_SAMPLE_SLOTS alive at a time)
        self._d_index = torch.empty((self.N_SAMPLE_SLOTS, nbytes), dtype=torch.uint8, device=dev)
        self._d_scal = torch.empty((self.N_SAMPLE_SLOTS, B * (A + 2)), dtype=torch.float32, device=dev)
        self._sample_gen = [0] * self.N_SAMPLE_SLOTS
        self._sample_slot = -1
        if self.dedup_frames:
            # (obs stacks | next_obs stacks) of a minibatch, contiguous: also one [2B] ring for ObsRef.pair
            self._mb_store = torch.zeros((self.N_SAMPLE_SLOTS, 2 * B * frame + 32), dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ writing
    def _stage_scalars(self, row, action, reward, done):
        A = self._n_act
        sc = row[self._sc_off + self._hdr:].view(np.float32)
        sc[:A] = np.asarray(action, dtype=np.float32).reshape(-1)
        sc[A] = float(reward)
        sc[A + 1] = float(not done)
        return sc

    def _next_add_slot(self):
        k = self._add_slot
        self._add_slot = (k + 1) % self._n_add
        if self._add_events[k] is not None:
            self._add_events[k].synchronize()
        return k

    def add(self, obs, action, reward, next_obs, done):
        """utils.py:120-128: store one transition at ``idx``.  The two frames and the scalars travel in one
        pinned block and one async copy; two kernels turn CHW into the ring's HWC, one row copy stores the
        scalars.  Nothing here waits for the GPU (a slot is reused only after its copy has executed)."""
        if self.dedup_frames:
            return self._add_dedup(obs, action, reward, next_obs, done)
        i = self.idx
        k = self._next_add_slot()
        fr = self._frame
        row = self._h_add_np[k]
        row[:fr] = np.asarray(obs, dtype=np.uint8).reshape(-1)
        row[fr:2 * fr] = np.asarray(next_obs, dtype=np.uint8).reshape(-1)
        sc = self._stage_scalars(row, action, reward, done)
        if self.device.type == "cuda":
            self._d_add.copy_(self._h_add[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._add_events[k] = ev
            ops.store_frame(self._d_add_frames[0], self.obses, i)
            ops.store_frame(self._d_add_frames[1], self.next_obses, i)
            self._sc[i].copy_(self._d_add_sc)
        else:  # host-side bookkeeping only (index logic tests); pixels are still stored, HWC
            c, h, w = self.obs_shape
            blk = self._h_add[k]
            self.obses[i] = blk[:fr].view(c, h, w).permute(1, 2, 0)
            self.next_obses[i] = blk[fr:2 * fr].view(c, h, w).permute(1, 2, 0)
            self._sc[i] = torch.from_numpy(sc.copy())
        self._advance(1)

    def _advance(self, n):
        new_idx = self.idx + n
        self.full = self.full or new_idx >= self.capacity
        self.idx = new_idx % self.capacity

    def _add_dedup(self, obs, action, reward, next_obs, done):
        """add() into the frame store: each of the 2k RGB frames of (obs, next_obs) is matched byte for byte
        against the recently stored ones (in a frame-stacked episode 2k-1 of them are), only the new ones are
        uploaded, and the transition records 2k frame ids."""
        c, h, w = self.obs_shape
        K, f3 = self._k, 3 * h * w
        i = self.idx
        st = self._store
        old = self._fid_h[i].reshape(-1)
        slot = self._next_add_slot()
        row = self._h_add_np[slot]
        ids, new = np.empty(2 * K, dtype=np.int32), []
        frames = np.concatenate([np.asarray(obs, dtype=np.uint8).reshape(K, 3, h, w),
                                 np.asarray(next_obs, dtype=np.uint8).reshape(K, 3, h, w)])
        # Room first, nothing touched yet: the frames this transition does not find in the store need a free slot
        # each (a frame repeated inside the transition is counted once per occurrence: an upper bound), and the
        # slots that the transition being overwritten (ring wrap) gives back count as free.  Failing here leaves the
        # store exactly as it was -- a caller that catches the MemoryError (add_batch / load loops) keeps a
        # consistent buffer.
        found = [st.lookup(frames[j])[0] for j in range(2 * K)]
        keeps = collections.Counter(f for f in found if f is not None)
        gives = collections.Counter(int(f) for f in old if f >= 0)
        freed = sum(1 for f, n in gives.items() if st.refs[f] - n + keeps.get(f, 0) == 0)
        if sum(f is None for f in found) > len(st.free) + freed:
            raise MemoryError("frame store exhausted: the observations handed to add() share fewer frames than a "
                              "frame-stacked episode does; raise frame_capacity or construct the ReplayBuffer with "
                              "dedup_frames=False")
        # 1. hold on to the frames that are already there, 2. let the overwritten transition go (frames it shares with
        #    the new one stay: held), 3. store the new frames -- in the slots step 2 may just have freed
        for f in found:
            if f is not None:
                st.refs[f] += 1
        for fid in old:
            if fid >= 0:
                st.release(int(fid))
        for j in range(2 * K):
            fid = found[j]
            if fid is None:
                fid, fp = st.lookup(frames[j])  # (an earlier frame of this transition may have just stored it)
                if fid is None:
                    fid = st.allocate(frames[j], fp)
                    row[len(new) * f3:(len(new) + 1) * f3] = frames[j].reshape(-1)
                    new.append(fid)
                st.refs[fid] += 1
            ids[j] = fid
        self._fid_h[i] = ids.reshape(2, K)
        row[self._sc_off:self._sc_off + self._hdr].view(np.int32)[:] = ids
        sc = self._stage_scalars(row, action, reward, done)
        if self.device.type == "cuda":
            n = len(new) * f3
            if n:
                self._d_add[:n].copy_(self._h_add[slot, :n], non_blocking=True)
            self._d_add[self._sc_off:].copy_(self._h_add[slot, self._sc_off:], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._add_events[slot] = ev
            for j, fid in enumerate(new):
                ops.store_frame(self._d_add[j * f3:(j + 1) * f3], self.frames, fid)
            self._fid[i].view(-1).copy_(self._d_add[self._sc_off:self._sc_off + self._hdr].view(torch.int32))
            self._sc[i].copy_(self._d_add_sc)
        else:  # host-side bookkeeping only
            for j, fid in enumerate(new):
                self.frames[fid] = torch.from_numpy(row[j * f3:(j + 1) * f3].reshape(3, h, w).copy()).permute(1, 2, 0)
            self._fid[i] = torch.from_numpy(ids.reshape(2, K).copy())
            self._sc[i] = torch.from_numpy(sc.copy())
        self._advance(1)

    def add_batch(self, obses, actions, rewards, next_obses, dones):
        """Bulk fill (benchmarks / buffer load): N transitions, observations as
        (N, C, H, W) uint8 arrays.  Same ring semantics as N add() calls."""
        n = len(obses)
        if self.dedup_frames:
            for j in range(n):
                self.add(obses[j], actions[j], float(np.asarray(rewards[j]).reshape(-1)[0]), next_obses[j],
                         bool(np.asarray(dones[j]).reshape(-1)[0]))
            return
        for s in range(0, n, 1024):
            e = min(n, s + 1024)
            m = e - s
            o = torch.from_numpy(np.ascontiguousarray(obses[s:e])).to(self.device).permute(0, 2, 3, 1)
            nx = torch.from_numpy(np.ascontiguousarray(next_obses[s:e])).to(self.device).permute(0, 2, 3, 1)
            slots = (torch.arange(m) + self.idx) % self.capacity
            slots_d = slots.to(self.device)
            self.obses[slots_d] = o
            self.next_obses[slots_d] = nx
            self.actions[slots_d] = torch.as_tensor(np.asarray(actions[s:e], dtype=np.float32), device=self.device)
            self.rewards[slots_d] = torch.as_tensor(np.asarray(rewards[s:e], dtype=np.float32).reshape(m, 1),
                                                    device=self.device)
            nd = 1.0 - np.asarray(dones[s:e], dtype=np.float32).reshape(m, 1)
            self.not_dones[slots_d] = torch.as_tensor(nd, device=self.device)
            self._advance(m)

    def frames_in_use(self):
        """(dedup_frames) RGB frames currently held / the store's capacity."""
        return self.frame_capacity - len(self._store.free), self.frame_capacity

    # ------------------------------------------------------------------ sampling
    def _is_crop(self):
        return isinstance(self.augmentor, augmentations.RandomCrop)

    def draw_indices(self):
        """Host RNG draws of sample_cpc, in the reference's order (utils.py:147 then
        augmentations.py:66-67 for obs, next_obs, pos).  Returns (idxs, offsets) with
        offsets an int32 array [6, B] = h1/w1 of obs, next_obs, pos (zeros when the
        augmentation is not RandomCrop)."""
        B = self.batch_size
        idxs = np.random.randint(0, self.capacity if self.full else self.idx, size=B)
        offs = np.zeros((6, B), dtype=np.int32)
        if self._is_crop():
            for j in range(3):
                h1, w1 = self.augmentor.draw_offsets(B)
                offs[2 * j], offs[2 * j + 1] = h1, w1
        elif not isinstance(self.augmentor, (augmentations.ColorJiggle, augmentations.NoisyCover)) and \
                type(self.augmentor) is not augmentations.IdentityAugmentation:
            raise NotImplementedError("unknown augmentation object: %r" % (self.augmentor,))
        return idxs, offs

    def _float_augmented(self, ring, idx, out=None):
        """One augmented float NHWC minibatch [B, H, W, C] from ``ring`` rows ``idx`` (None: rows 0..B-1)
        (utils.py:168-182 branch: the torch/kornia augmentations)."""
        B = self.batch_size
        c, h, w = self.obs_shape
        if out is None:
            out = torch.empty((B, h, w, c), dtype=torch.float32, device=self.device)
        aug = self.augmentor
        if isinstance(aug, augmentations.ColorJiggle):
            params, order = aug.draw_params(B * (c // 3))
            # one pinned staging block, one asynchronous copy: a `.to(device)` of a pageable tensor makes the host wait
            # until the stream has drained (two of them per call left ~60 us of idle GPU around every jitter launch)
            n4 = params.numel()
            cuda = self.device.type == "cuda"
            stage = torch.empty(n4 + 4, dtype=torch.int32, pin_memory=cuda)
            stage[:n4] = params.reshape(-1).view(torch.int32)
            stage[n4:] = order
            d = stage.to(self.device, non_blocking=True)
            ops.color_jiggle(ring, idx, d[:n4].view(torch.float32).view(params.shape), d[n4:], B, out)
        elif isinstance(aug, augmentations.NoisyCover):
            colors = aug.draw_colors()
            noise = torch.randn((B, h, w, c), device=self.device) * aug.std
            ops.noisy_cover(ring, idx, noise, colors, aug.top, aug.bottom, B, out)
        else:
            ops.gather_nhwc(ring, idx, B, out)
        return out

    def _is_float_aug(self):
        return isinstance(self.augmentor, (augmentations.ColorJiggle, augmentations.NoisyCover))

    def _fill_index_block(self, host, idxs, offs):
        """A minibatch's indices and crop offsets in the layout the kernels read:
        idx [B] | idx + capacity [B] (the same transitions in the next_obs half of the double ring) | h1 of obs,
        next_obs, pos | w1 of obs, next_obs, pos -- so that (obs, next_obs) is ONE run of 2B frame indices, 2B row
        offsets and 2B column offsets."""
        B = self.batch_size
        i64 = host[:2 * B * 8].view(torch.int64)
        i64[:B].copy_(torch.from_numpy(np.ascontiguousarray(idxs, dtype=np.int64)))
        i64[B:].copy_(i64[:B] + self.capacity)
        o32 = host[2 * B * 8:2 * B * 8 + 6 * B * 4].view(torch.int32).view(6, B)
        offs = np.ascontiguousarray(offs, dtype=np.int32)
        o32.copy_(torch.from_numpy(np.ascontiguousarray(offs[[0, 2, 4, 1, 3, 5]])))

    def _index_views(self, dst):
        """(idx [B] int64, the six offset rows, the 2B-long (idx, h1, w1) views) of a device index block."""
        B = self.batch_size
        d64 = dst[:2 * B * 8].view(torch.int64)
        d32 = dst[2 * B * 8:2 * B * 8 + 6 * B * 4].view(torch.int32)
        # off[2j] / off[2j+1] = h1 / w1 of tensor j (obs, next_obs, pos); pair = the 2B-long views
        off = [d32[(j // 2 + 3 * (j % 2)) * B:(j // 2 + 3 * (j % 2) + 1) * B] for j in range(6)]
        return d64[:B], off, (d64, d32[:2 * B], d32[3 * B:5 * B])

    def _upload_indices(self, idxs, offs):
        """Copy a minibatch's indices and crop offsets into the next device sample slot; returns the slot's
        (guard, idx view [B] int64, offsets view [6, B] int32)."""
        B = self.batch_size
        u, every = self._slot_use, self.EVENT_EVERY
        self._slot_use = u + 1
        k = u % self._n_slots
        if u >= self._n_slots:
            e = (u - self._n_slots) // every
            ev = self._slot_events.get(e)
            if ev is not None:
                ev.synchronize()
            for old in [i for i in self._slot_events if i < e]:
                del self._slot_events[old]
        host = self._h_index[k]
        self._fill_index_block(host, idxs, offs)
        s = self._sample_slot = (self._sample_slot + 1) % self.N_SAMPLE_SLOTS
        self._sample_gen[s] += 1
        dst = self._d_index[s]
        self._staged = self._h_index_dev is not None
        if self._staged:  # index block and the transitions' scalars in one launch, the block read from the pinned slot
            B, A = self.batch_size, self._n_act
            buf = self._d_scal[s]
            ops.sample_stage(self._h_index_dev[k], dst, host.numel(), self._sc, B, A, buf[:B * A], buf[B * A:B * A + B],
                             buf[B * A + B:])
        else:
            dst.copy_(host, non_blocking=True)
        if self.device.type == "cuda" and u % every == every - 1:
            ev = torch.cuda.Event()
            ev.record()
            self._slot_events[u // every] = ev
        guard = (self._sample_gen, s, self._sample_gen[s])
        d_idx, off, self._pair_views = self._index_views(dst)
        return guard, d_idx, off

    def _scalars(self, d_idx):
        """actions [B, ...], rewards [B, 1], not_dones [B, 1] of the sampled transitions (utils.py:159-166): one
        gather kernel into this sample slot's buffer (valid as long as the slot's pixel handles are)."""
        B, A = self.batch_size, self._n_act
        buf = self._d_scal[self._sample_slot]
        act, rew, nd = buf[:B * A].view((B,) + tuple(self.actions.shape[1:])), buf[B * A:B * A + B].view(B, 1), \
            buf[B * A + B:].view(B, 1)
        if self._staged:  # gathered by the launch that staged the indices (_upload_indices)
            self._staged = False
        elif self.device.type == "cuda" or _lib_tracing():
            ops.gather_transition_scalars(self._sc, d_idx, B, A, act, rew, nd)
        return act, rew, nd

    def _require_cuda(self):
        from . import _lib
        if self.device.type != "cuda" and _lib._trace_hook is None:
            raise RuntimeError("sampling pixels needs the HIP device: curla_amd has no CPU fallback for the learner path")

    def _sources(self, d_idx):
        """(obs ring, next_obs ring, row index tensor or None) the loaders read a minibatch from.  Plain storage:
        the two rings, indexed by the sampled slots.  De-duplicated storage: the k frames of every sampled stack
        are first assembled into this sample slot's [B][H][W][3k] uint8 buffers (one gather kernel per tensor)."""
        if not self.dedup_frames:
            return self.obses, self.next_obses, d_idx
        B = self.batch_size
        c, h, w = self.obs_shape
        mb = self._mb_store[self._sample_slot]
        both = mb[:2 * B * self._frame].view(2 * B, h, w, c)
        views = [both[:B], both[B:]]
        for j in range(2):
            ops.gather_stacks(self.frames, self._fid[:, j, :], d_idx, B, views[j])
        self._mb_both = both
        return views[0], views[1], None

    # ---- dedicated sample slots of captured update graphs (CurlSacAgent.enable_update_graphs) ---------------------
    # A captured graph replays the SAME pointers: its minibatch block lives in its own pinned host slot and its own
    # device block, never in the rotating ones above.  Behind the indices the block carries GRAPH_TAIL bytes of per-update
    # control values (RNG stream positions, Adam step factors) that the graph's kernels read from the device copy.
    GRAPH_TAIL = 80  # u64[4] (seed, critic-noise offset, seed, actor-noise offset) | f64[2] log_alpha | f32[8] four Adams

    def graph_supported(self):
        """Graph replay covers the uint8-ring minibatches (RandomCrop / identity, plain storage, one allocation for both
        rings, pinned index slots read in place); the float augmentations stage their parameters through per-call
        pinned blocks and stay eager."""
        return (self.device.type == "cuda" and not self._is_float_aug() and not self.dedup_frames
                and self._both is not None and self._h_index_dev is not None)

    def graph_block(self, slot):
        if not hasattr(self, "_graph_blocks"):
            self._graph_blocks = {}
        g = self._graph_blocks.get(slot)
        if g is None:
            B, A = self.batch_size, self._n_act
            nb = self._h_index.shape[1] + self.GRAPH_TAIL
            host = torch.zeros(nb, dtype=torch.uint8, pin_memory=True)
            g = dict(host=host, host_dev=ops.host_device_pointer(host), dev=torch.zeros(nb, dtype=torch.uint8, device=self.device),
                     scal=torch.empty(B * (A + 2), dtype=torch.float32, device=self.device), event=None,
                     tail=self._h_index.shape[1])
            self._graph_blocks[slot] = g
        return g

    def graph_write(self, slot, idxs, offs, tail):
        """Host side of one graphed update: the minibatch's indices / offsets and the control tail (80 bytes) into the
        slot's pinned block -- after the previous replay that reads this block has finished."""
        g = self.graph_block(slot)
        if g["event"] is not None:
            g["event"].synchronize()
        self._fill_index_block(g["host"], idxs, offs)
        g["host"][g["tail"]:].copy_(torch.from_numpy(np.frombuffer(bytearray(tail), dtype=np.uint8)))
        return g

    def graph_refs(self, slot):
        """Device side, called while the graph is being captured: the staging launch (pinned block -> device block +
        the transitions' scalars) and the sample_cpc 6-tuple with handles into the slot's device block."""
        g = self.graph_block(slot)
        B, A = self.batch_size, self._n_act
        buf = g["scal"]
        ops.sample_stage(g["host_dev"], g["dev"], g["host"].numel(), self._sc, B, A, buf[:B * A], buf[B * A:B * A + B],
                         buf[B * A + B:])
        d_idx, off, (idx2, h2, w2) = self._index_views(g["dev"])
        crop = tuple(self.augmentor.output_shape)
        both = self._both
        obses = ops.ObsRef.from_ring(both, idx2[:B], off[0], off[1], B, crop, None)
        next_obses = ops.ObsRef.from_ring(both, idx2[B:], off[2], off[3], B, crop, None)
        pos = ops.ObsRef.from_ring(both, idx2[:B], off[4], off[5], B, crop, None)
        obses.pair = (ops.ObsRef.from_ring(both, idx2, h2, w2, 2 * B, crop, None), next_obses)
        act, rew, nd = buf[:B * A].view((B,) + tuple(self.actions.shape[1:])), buf[B * A:B * A + B].view(B, 1), \
            buf[B * A + B:].view(B, 1)
        return obses, act, rew, next_obses, nd, dict(obs_anchor=obses, obs_pos=pos, time_anchor=None, time_pos=None)

    def sample_cpc_refs(self, indices=None):
        """The fused form of sample_cpc: same 6-tuple, but obs / next_obs / pos are
        ``ObsRef`` handles (ring + indices + crop offsets) consumed directly by the
        first conv kernel.  ``indices=(idxs, offs)`` injects pre-drawn indices (tests, DP).
        The handles of one call stay valid until N_SAMPLE_SLOTS further samples have been drawn (using an older
        one raises)."""
        self._require_cuda()
        idxs, offs = indices if indices is not None else self.draw_indices()
        guard, d_idx, off = self._upload_indices(idxs, offs)
        B = self.batch_size
        crop = tuple(self.augmentor.output_shape)
        ring_o, ring_n, rows = self._sources(d_idx)
        if self._is_float_aug():
            # obs, next_obs and pos (= a copy of obs) are augmented independently (utils.py:173-182); obs and
            # next_obs are written into the two halves of one [2B] tensor (ObsRef.pair, see below)
            c, h, w = self.obs_shape
            both = torch.empty((2 * B, h, w, c), dtype=torch.float32, device=self.device)
            obses = ops.ObsRef.from_nhwc(self._float_augmented(ring_o, rows, out=both[:B]))
            next_obses = ops.ObsRef.from_nhwc(self._float_augmented(ring_n, rows, out=both[B:]))
            pos = ops.ObsRef.from_nhwc(self._float_augmented(ring_o, rows))
            obses.pair = (ops.ObsRef.from_nhwc(both), next_obses)
        else:
            idx2, h2, w2 = self._pair_views
            both = self._mb_both if self.dedup_frames else self._both
            if both is None:  # (rings in two allocations: second half would not start on a dword)
                obses = ops.ObsRef.from_ring(ring_o, rows, off[0], off[1], B, crop, guard)
                next_obses = ops.ObsRef.from_ring(ring_n, rows, off[2], off[3], B, crop, guard)
                pos = ops.ObsRef.from_ring(ring_o, rows, off[4], off[5], B, crop, guard)
            else:
                # every handle indexes the ONE ring that holds obs frames then next_obs frames, so that any two of
                # them can share a first-layer launch (ops.conv1_fwd2), and (obs | next_obs) is itself a handle of
                # 2B frames: the critic phase runs both through the online convs in one launch per layer
                # (curl_sac.py:350-358)
                if self.dedup_frames:
                    if getattr(self, "_ar2", None) is None:
                        self._ar2 = torch.arange(2 * B, device=self.device, dtype=torch.int64)
                    idx2 = self._ar2
                obses = ops.ObsRef.from_ring(both, idx2[:B], off[0], off[1], B, crop, guard)
                next_obses = ops.ObsRef.from_ring(both, idx2[B:], off[2], off[3], B, crop, guard)
                pos = ops.ObsRef.from_ring(both, idx2[:B], off[4], off[5], B, crop, guard)
                obses.pair = (ops.ObsRef.from_ring(both, idx2, h2, w2, 2 * B, crop, guard), next_obses)
        actions, rewards, not_dones = self._scalars(d_idx)
        cpc_kwargs = dict(obs_anchor=obses, obs_pos=pos, time_anchor=None, time_pos=None)
        return obses, actions, rewards, next_obses, not_dones, cpc_kwargs

    def sample_cpc(self, indices=None):
        """utils.py:144-187 with the reference's return types: float32 NCHW tensors
        in [0,255] on the device (materialised by one crop kernel per tensor)."""
        self._require_cuda()
        idxs, offs = indices if indices is not None else self.draw_indices()
        _, d_idx, off = self._upload_indices(idxs, offs)
        B = self.batch_size
        c = self.obs_shape[0]
        oh, ow = self.augmentor.output_shape
        ring_o, ring_n, rows = self._sources(d_idx)
        outs = []
        for ring, j in ((ring_o, 0), (ring_n, 1), (ring_o, 2)):
            t = torch.empty((B, c, oh, ow), dtype=torch.float32, device=self.device)
            if self._is_float_aug():
                ops.nhwc_to_nchw(self._float_augmented(ring, rows), t)
            else:
                ops.crop_nchw(ring, rows, off[2 * j], off[2 * j + 1], B, (oh, ow), out_f32=t)
            outs.append(t)
        obses, next_obses, pos = outs
        actions, rewards, not_dones = (t.clone() for t in self._scalars(d_idx))  # fresh tensors, like the reference's
        cpc_kwargs = dict(obs_anchor=obses, obs_pos=pos, time_anchor=None, time_pos=None)
        return obses, actions, rewards, next_obses, not_dones, cpc_kwargs

    # ------------------------------------------------------------------ persistence
    def stacks(self, lo, hi, which=0):
        """Transitions [lo, hi) as the reference stores them: a (hi-lo, C, H, W) uint8 NumPy array of obs
        (which=0) or next_obs (which=1) stacks."""
        c, h, w = self.obs_shape
        n = hi - lo
        if n <= 0:
            return np.empty((0, c, h, w), dtype=np.uint8)
        if self.dedup_frames:
            rows = torch.arange(lo, hi, device=self.device, dtype=torch.int64)
            buf = torch.zeros(n * self._frame + 32, dtype=torch.uint8, device=self.device)
            out = buf[:n * self._frame].view(n, h, w, c)
            ops.gather_stacks(self.frames, self._fid[:, which, :], rows, n, out)
            ring = out
        else:
            ring = (self.obses, self.next_obses)[which][lo:hi]
        return ring.permute(0, 3, 1, 2).contiguous().cpu().numpy()

    def save(self, save_dir):
        """utils.py:189-202: the transitions added since the last call go to one file ``{start}_{end}.pt`` whose
        payload is the reference's (five arrays, observation stacks as CHW uint8)."""
        lo, hi = self.last_save, self.idx
        if lo == hi:
            return
        payload = [self.stacks(lo, hi, 0), self.stacks(lo, hi, 1)]
        payload += [t[lo:hi].cpu().numpy() for t in (self.actions, self.rewards, self.not_dones)]
        self.last_save = hi
        torch.save(payload, os.path.join(save_dir, '%d_%d.pt' % (lo, hi)))

    def load(self, save_dir):
        """utils.py:204-216: read the ``{start}_{end}.pt`` files of ``save_dir`` back in ascending order of
        ``start``; each file must continue where the previous one ended."""
        def span(name):
            lo, hi = os.path.splitext(name)[0].split('_')
            return int(lo), int(hi)

        for name in sorted(os.listdir(save_dir), key=lambda n: span(n)[0]):
            lo, hi = span(name)
            if lo != self.idx:
                raise AssertionError("chunk %s does not continue the buffer at index %d" % (name, self.idx))
            obs, nxt, act, rew, nd = torch.load(os.path.join(save_dir, name), weights_only=False)
            if self.dedup_frames:
                self.add_batch(obs, act, rew, nxt, 1.0 - np.asarray(nd))
                self.idx = hi  # (the reference's load does not wrap either)
                continue
            to_ring = lambda a: torch.as_tensor(a).to(self.device).permute(0, 2, 3, 1)  # noqa: E731  CHW -> HWC
            self.obses[lo:hi] = to_ring(obs)
            self.next_obses[lo:hi] = to_ring(nxt)
            for dst, src in ((self.actions, act), (self.rewards, rew), (self.not_dones, nd)):
                dst[lo:hi] = torch.as_tensor(src).to(self.device)
            self.idx = hi

    def __len__(self):
        return self.capacity
