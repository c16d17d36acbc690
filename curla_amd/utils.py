"""Replay buffer and small helpers of the learner path (reference: utils.py).

``ReplayBuffer`` keeps the reference's constructor, ``add``, ``sample_cpc``,
``save``/``load``, ``idx``/``full`` -- but the ring lives in HBM (uint8, NHWC so
a cropped row is one contiguous run) and sampling hands the learner *references*
(frame indices + crop offsets) instead of materialised float tensors: the
reference's 3 x B x C x H x W float32 host->device copy per update (utils.py:161-166)
disappears.  Index and crop-offset draws stay on the host in NumPy's legacy
global stream, in the reference's order, so a seeded run samples the same
transitions and windows (bit-exact).
"""
import os
import random

import numpy as np
import torch

from . import augmentations, ops


class eval_mode(object):
    """utils.py:21-34."""

    def __init__(self, *models):
        self.models = models

    def __enter__(self):
        self.prev_states = []
        for model in self.models:
            self.prev_states.append(model.training)
            model.train(False)

    def __exit__(self, *args):
        for model, state in zip(self.models, self.prev_states):
            model.train(state)
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
    try:
        os.mkdir(dir_path)
    except OSError:
        print('Unable to create directory ' + dir_path)
    return dir_path


class ReplayBuffer(object):
    """Buffer to store environment transitions (utils.py:80-236), HBM-resident."""

    def __init__(self, obs_shape, action_shape, capacity, batch_size, device, augmentor, transform=None):
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
        total_bytes = 2 * capacity * frame + capacity * (4 * int(np.prod(action_shape)) + 8)
        if self.device.type == "cuda":
            free, _ = torch.cuda.mem_get_info(self.device)
            if total_bytes > free:
                raise ValueError('Replay buffer size exceeds available memory')  # utils.py:112-113
        # ring storage: NHWC uint8 frames (+32 B slack: the aligning loader reads whole 16-byte runs plus one dword)
        self._obs_store = torch.zeros(capacity * frame + 32, dtype=torch.uint8, device=self.device)
        self._next_store = torch.zeros(capacity * frame + 32, dtype=torch.uint8, device=self.device)
        self.obses = self._obs_store[:capacity * frame].view(capacity, h, w, c)
        self.next_obses = self._next_store[:capacity * frame].view(capacity, h, w, c)
        # action | reward | not_done of a transition sit in one row, so add() writes them with one small copy;
        # the three reference attributes are column views of it
        A = int(np.prod(action_shape))
        self._n_act = A
        self._sc = torch.empty((capacity, A + 2), dtype=torch.float32, device=self.device)
        self.actions = self._sc[:, :A].unflatten(1, tuple(action_shape)) if len(action_shape) != 1 else self._sc[:, :A]
        self.rewards = self._sc[:, A:A + 1]
        self.not_dones = self._sc[:, A + 1:A + 2]
        self.idx = 0
        self.last_save = 0
        self.full = False
        # staging: pinned host rows for add(), static device index buffers for sampling
        pin = self.device.type == "cuda"
        # add(): one pinned block per slot = [obs frame | next_obs frame | pad | action, reward, not_done], a few
        # slots guarded by events so that add() never waits for the GPU; one device block receives the copy
        self._frame = frame
        self._sc_off = (2 * frame + 15) & ~15
        blk = self._sc_off + 4 * (A + 2)
        self._n_add, self._add_slot = 4, 0
        self._h_add = torch.empty((self._n_add, blk), dtype=torch.uint8, pin_memory=pin)
        self._h_add_np = self._h_add.numpy()
        self._add_events = [None] * self._n_add
        self._d_add = torch.empty(blk, dtype=torch.uint8, device=self.device)
        self._d_add_frames = self._d_add[:2 * frame].view(2, frame)
        self._d_add_sc = self._d_add[self._sc_off:].view(torch.float32)
        B = batch_size
        # the host may run several updates ahead of the GPU: a small ring of pinned slots, each guarded by an
        # event, keeps an index upload's source intact until its async copy has executed
        self._n_slots, self._slot = 8, 0
        self._h_index = torch.empty((self._n_slots, B * 8 + B * 4 * 6), dtype=torch.uint8, pin_memory=pin)
        self._slot_events = [None] * self._n_slots
        self._d_index = torch.empty(B * 8 + B * 4 * 6, dtype=torch.uint8, device=self.device)
        self._d_idx = self._d_index[:B * 8].view(torch.int64)
        self._d_off = self._d_index[B * 8:].view(torch.int32).view(6, B)

    # ------------------------------------------------------------------ writing
    def add(self, obs, action, reward, next_obs, done):
        """utils.py:120-128: store one transition at ``idx``.  The two frames and the scalars travel in one
        pinned block and one async copy; two kernels turn CHW into the ring's HWC, one row copy stores the
        scalars.  Nothing here waits for the GPU (a slot is reused only after its copy has executed)."""
        i = self.idx
        k = self._add_slot
        self._add_slot = (k + 1) % self._n_add
        if self._add_events[k] is not None:
            self._add_events[k].synchronize()
        fr, A = self._frame, self._n_act
        row = self._h_add_np[k]
        row[:fr] = np.asarray(obs, dtype=np.uint8).reshape(-1)
        row[fr:2 * fr] = np.asarray(next_obs, dtype=np.uint8).reshape(-1)
        sc = row[self._sc_off:].view(np.float32)
        sc[:A] = np.asarray(action, dtype=np.float32).reshape(-1)
        sc[A] = float(reward)
        sc[A + 1] = float(not done)
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
        self.idx = (self.idx + 1) % self.capacity
        self.full = self.full or self.idx == 0

    def add_batch(self, obses, actions, rewards, next_obses, dones):
        """Bulk fill (benchmarks / buffer load): N transitions, observations as
        (N, C, H, W) uint8 arrays.  Same ring semantics as N add() calls."""
        n = len(obses)
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
            new_idx = self.idx + m
            self.full = self.full or new_idx >= self.capacity
            self.idx = new_idx % self.capacity

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

    def _float_augmented(self, ring):
        """One augmented float NHWC minibatch [B, H, W, C] from ``ring`` at the uploaded indices
        (utils.py:168-182 branch: the torch/kornia augmentations)."""
        B = self.batch_size
        c, h, w = self.obs_shape
        out = torch.empty((B, h, w, c), dtype=torch.float32, device=self.device)
        aug = self.augmentor
        if isinstance(aug, augmentations.ColorJiggle):
            params, order = aug.draw_params(B * (c // 3))
            ops.color_jiggle(ring, self._d_idx, params.to(self.device), order.to(self.device), B, out)
        elif isinstance(aug, augmentations.NoisyCover):
            colors = aug.draw_colors()
            noise = torch.randn((B, h, w, c), device=self.device) * aug.std
            ops.noisy_cover(ring, self._d_idx, noise, colors, aug.top, aug.bottom, B, out)
        else:
            ops.gather_nhwc(ring, self._d_idx, B, out)
        return out

    def _is_float_aug(self):
        return isinstance(self.augmentor, (augmentations.ColorJiggle, augmentations.NoisyCover))

    def _upload_indices(self, idxs, offs):
        B = self.batch_size
        k = self._slot
        self._slot = (k + 1) % self._n_slots
        if self._slot_events[k] is not None:
            self._slot_events[k].synchronize()
        host = self._h_index[k]
        host[:B * 8].view(torch.int64).copy_(torch.from_numpy(np.ascontiguousarray(idxs, dtype=np.int64)))
        host[B * 8:].view(torch.int32).view(6, B).copy_(torch.from_numpy(np.ascontiguousarray(offs, dtype=np.int32)))
        self._d_index.copy_(host, non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._slot_events[k] = ev

    def _scalars(self):
        return (self.actions.index_select(0, self._d_idx), self.rewards.index_select(0, self._d_idx),
                self.not_dones.index_select(0, self._d_idx))

    def _require_cuda(self):
        from . import _lib
        if self.device.type != "cuda" and _lib._trace_hook is None:
            raise RuntimeError("sampling pixels needs the HIP device: curla_amd has no CPU fallback for the learner path")

    def sample_cpc_refs(self, indices=None):
        """The fused form of sample_cpc: same 6-tuple, but obs / next_obs / pos are
        ``ObsRef`` handles (ring + indices + crop offsets) consumed directly by the
        first conv kernel.  ``indices=(idxs, offs)`` injects pre-drawn indices (tests, DP)."""
        self._require_cuda()
        idxs, offs = indices if indices is not None else self.draw_indices()
        self._upload_indices(idxs, offs)
        B = self.batch_size
        crop = tuple(self.augmentor.output_shape)
        off = self._d_off
        if self._is_float_aug():
            # obs, next_obs and pos (= a copy of obs) are augmented independently (utils.py:173-182)
            obses = ops.ObsRef.from_nhwc(self._float_augmented(self.obses))
            next_obses = ops.ObsRef.from_nhwc(self._float_augmented(self.next_obses))
            pos = ops.ObsRef.from_nhwc(self._float_augmented(self.obses))
        else:
            obses = ops.ObsRef.from_ring(self.obses, self._d_idx, off[0], off[1], B, crop)
            next_obses = ops.ObsRef.from_ring(self.next_obses, self._d_idx, off[2], off[3], B, crop)
            pos = ops.ObsRef.from_ring(self.obses, self._d_idx, off[4], off[5], B, crop)
        actions, rewards, not_dones = self._scalars()
        cpc_kwargs = dict(obs_anchor=obses, obs_pos=pos, time_anchor=None, time_pos=None)
        return obses, actions, rewards, next_obses, not_dones, cpc_kwargs

    def sample_cpc(self, indices=None):
        """utils.py:144-187 with the reference's return types: float32 NCHW tensors
        in [0,255] on the device (materialised by one crop kernel per tensor)."""
        self._require_cuda()
        idxs, offs = indices if indices is not None else self.draw_indices()
        self._upload_indices(idxs, offs)
        B = self.batch_size
        c = self.obs_shape[0]
        oh, ow = self.augmentor.output_shape
        off = self._d_off
        outs = []
        for ring, j in ((self.obses, 0), (self.next_obses, 1), (self.obses, 2)):
            t = torch.empty((B, c, oh, ow), dtype=torch.float32, device=self.device)
            if self._is_float_aug():
                ops.nhwc_to_nchw(self._float_augmented(ring), t)
            else:
                ops.crop_nchw(ring, self._d_idx, off[2 * j], off[2 * j + 1], B, (oh, ow), out_f32=t)
            outs.append(t)
        obses, next_obses, pos = outs
        actions, rewards, not_dones = self._scalars()
        cpc_kwargs = dict(obs_anchor=obses, obs_pos=pos, time_anchor=None, time_pos=None)
        return obses, actions, rewards, next_obses, not_dones, cpc_kwargs

    # ------------------------------------------------------------------ persistence
    def _chw(self, ring, lo, hi):
        return ring[lo:hi].permute(0, 3, 1, 2).contiguous().cpu().numpy()

    def save(self, save_dir):
        """utils.py:189-202: incremental chunk ``{start}_{end}.pt`` in the reference's
        payload format (CHW uint8 NumPy arrays)."""
        if self.idx == self.last_save:
            return
        path = os.path.join(save_dir, '%d_%d.pt' % (self.last_save, self.idx))
        lo, hi = self.last_save, self.idx
        payload = [self._chw(self.obses, lo, hi), self._chw(self.next_obses, lo, hi),
                   self.actions[lo:hi].cpu().numpy(), self.rewards[lo:hi].cpu().numpy(),
                   self.not_dones[lo:hi].cpu().numpy()]
        self.last_save = self.idx
        torch.save(payload, path)

    def load(self, save_dir):
        """utils.py:204-216."""
        chunks = os.listdir(save_dir)
        chucks = sorted(chunks, key=lambda x: int(x.split('_')[0]))
        for chunk in chucks:
            start, end = [int(x) for x in chunk.split('.')[0].split('_')]
            path = os.path.join(save_dir, chunk)
            payload = torch.load(path, weights_only=False)
            assert self.idx == start
            dev = self.device
            self.obses[start:end] = torch.as_tensor(payload[0]).to(dev).permute(0, 2, 3, 1)
            self.next_obses[start:end] = torch.as_tensor(payload[1]).to(dev).permute(0, 2, 3, 1)
            self.actions[start:end] = torch.as_tensor(payload[2]).to(dev)
            self.rewards[start:end] = torch.as_tensor(payload[3]).to(dev)
            self.not_dones[start:end] = torch.as_tensor(payload[4]).to(dev)
            self.idx = end

    def __len__(self):
        return self.capacity
