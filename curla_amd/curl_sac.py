"""CURL + SAC learner on MI355X: Actor / QFunction / Critic / CURL /
CurlSacAgent with the reference's API (curl_sac.py:57-465).

Host code is Python on PyTorch-ROCm: parameters are ``nn.Parameter`` views into
flat HBM buffers, the five optimizers are ``torch.optim.Adam`` objects (four of
them ``FlatAdam``: the same optimizer with its step as one flat HIP launch) --
the reference's division of labour.  Everything between "minibatch indices" and
".grad is filled in" runs as hand-written HIP kernels through the C ABI
(include/curla_hip.h); there is no autograd graph and no PyTorch fallback.

Schedule of one ``update()`` (reference curl_sac.py:426-451), with the sharing
the reference's autograd cannot do:
  critic phase : conv(next_obs; theta) -> actor head;  conv(next_obs; xi) -> target Q;
                 conv(obs; theta) -> twin Q -> loss -> explicit backward -> Adam
  actor phase  : conv(obs; theta') ONCE, reused by actor.fc, critic.fc and (below) CURL;
                 only live gradients are computed (actor fc/ln/trunk, log_alpha)
  soft update  : two flat lerps (Q block with critic_tau, encoder block with encoder_tau)
  cpc phase    : anchor features reused from the actor phase on even steps;
                 conv(pos; xi') -> bilinear logits (MFMA) -> CE -> backward -> 2x Adam
=> 5 conv-stack forwards + 2 backwards per update instead of the reference's 7 + 2.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .encoder import CNNEncoder
from .ops import ObsRef
from .optim import FlatAdam

LOG_FREQ = 25_000


def weight_init(m):
    """The reference's initialisation (curl_sac.py:38-54), applied with ``module.apply``: Linear weights orthogonal
    with zero bias; Conv2d / ConvTranspose2d "delta-orthogonal" (arXiv:1806.05393) -- every tap zero except the
    centre one, which is an orthogonal (out x in) matrix with the ReLU gain -- and zero bias.  The draws come from
    torch's global generator in module order, so a seeded construction reproduces the reference's parameters."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.zero_()
        return
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        kh, kw = m.weight.shape[2:]
        assert kh == kw, "square kernels only"
        m.weight.data.zero_()
        if m.bias is not None:
            m.bias.data.zero_()
        centre = kh // 2
        nn.init.orthogonal_(m.weight.data[:, :, centre, centre], nn.init.calculate_gain('relu'))


def _as_ref(obs):
    if isinstance(obs, ObsRef):
        return obs
    return ObsRef.from_tensor(obs.contiguous().float())


class _Mlp:
    """Pointers of a 3-layer MLP (or of the first of two identically laid-out
    twins, ``stride`` floats apart)."""

    def __init__(self, seq, stride=0, grads=False):
        lin = [seq[0], seq[2], seq[4]]
        pick = (lambda p: p.grad) if grads else (lambda p: p)
        self.W = [pick(layer.weight) for layer in lin]
        self.b = [pick(layer.bias) for layer in lin]
        self.stride = stride


def _mlp_fwd(x, sx, P, nb, B, din, H, dout, h1, h2, out, relu_out=0, outer=None, head=None):
    """``outer`` = (n2, sW2): n2 such MLP groups in the same launches, their parameters sW2 floats apart (the target
    critic's and the critic's twins); x [n2][B, din] and h1 / h2 / out [n2][nb][B, .] then.
    ``head`` = (noise, lo, hi, outputs): the MLP is the actor trunk and the policy head (ops.actor_head_fwd with these
    arguments) follows its last layer -- in the same launch when the layer is small enough."""
    s = P.stride
    o0 = o1 = o2 = None
    if outer is not None:
        n2, sW2 = outer
        o0, o1, o2 = (n2, B * din, sW2, nb * B * H), (n2, nb * B * H, sW2, nb * B * H), (n2, nb * B * H, sW2, nb * B * dout)
    ops.linear_fwd(x, sx, P.W[0], s, P.b[0], s, h1, B * H, B, H, din, nb, relu=1, outer=o0)
    ops.linear_fwd(h1, B * H, P.W[1], s, P.b[1], s, h2, B * H, B, H, H, nb, relu=1, outer=o1)
    small = dout <= ops.MLP_OUT_MAX and H % 4 == 0 and not relu_out
    if head is not None and small and nb == 1 and outer is None:
        noise, lo, hi, outs = head
        ops.mlp_out_head_fwd(h2, P.W[2], P.b[2], out, noise, B, dout // 2, H, lo, hi, **outs)
        return
    if small:  # a handful of outputs: row dot products, not a GEMM
        ops.mlp_out_fwd(h2, B * H, P.W[2], s, P.b[2], s, out, B * dout, B, dout, H, nb, outer=o2)
    else:
        ops.linear_fwd(h2, B * H, P.W[2], s, P.b[2], s, out, B * dout, B, dout, H, nb, relu=relu_out, outer=o2)
    if head is not None:
        noise, lo, hi, outs = head
        ops.actor_head_fwd(out, noise, B, dout // 2, lo, hi, **outs)


def _mlp_bwd(x, sx, P, G, nb, B, din, H, dout, h1, h2, dy, dh2, dh1, dx, loss=None):
    """Backward of _mlp_fwd.  G (parameter gradients) and dx are optional.
    ``loss`` = (fields, launch): the output gradient dy is that of a loss over the twin Q values -- ``fields`` for
    ops.mlp_out_bwd_loss, which computes it inside the last layer's backward launch; ``launch()`` runs the loss kernel
    on its own where that form does not apply."""
    s = P.stride
    fused_loss = loss is not None and dout == 1 and nb == 2
    if loss is not None and not fused_loss:
        loss[1]()
    # the three bias gradients (column sums of dy, dh2, dh1) ride in the launches that walk those matrices anyway --
    # the last layer's backward pass and the first layer's weight-gradient product -- where both take their small forms
    fold = G is not None and dout <= ops.MLP_OUT_MAX and ops.linear_dw_folds_bias(B, H, din, nb)
    if fused_loss:
        ops.mlp_out_bwd_loss(loss[0], h2, B * H, P.W[2], s, dh2, B * H, G.W[2] if G is not None else None, s, B, H,
                             db_out=G.b[2] if fold else None, db_hidden=G.b[1] if fold else None, sdb=s)
    elif dout <= ops.MLP_OUT_MAX:  # last layer: data and weight gradient in one pass over h2
        ops.mlp_out_bwd(dy, B * dout, h2, B * H, P.W[2], s, dh2, B * H, G.W[2] if G is not None else None, s, B, dout,
                        H, nb, db_out=G.b[2] if fold else None, db_hidden=G.b[1] if fold else None, sdb=s)
    else:
        if G is not None:
            ops.linear_dw(dy, B * dout, h2, B * H, G.W[2], s, B, dout, H, nb)
        ops.linear_dx(dy, B * dout, P.W[2], s, dh2, B * H, B, dout, H, nb, mask=h2, smask=B * H)
    # a layer's weight and data gradient are independent products over the same dy: one launch for the pair
    if G is not None:
        ops.linear_bwd(dh2, B * H, h1, B * H, P.W[1], s, G.W[1], s, dh1, B * H, B, H, H, nb, mask=h1, smask=B * H)
    else:
        ops.linear_dx(dh2, B * H, P.W[1], s, dh1, B * H, B, H, H, nb, mask=h1, smask=B * H)
    if G is not None and dx is not None:
        ops.linear_bwd(dh1, B * H, x, sx, P.W[0], s, G.W[0], s, dx, B * din, B, H, din, nb,
                       db=G.b[0] if fold else None, sdb=s)
    elif G is not None:
        ops.linear_dw(dh1, B * H, x, sx, G.W[0], s, B, H, din, nb, colsum=G.b[0] if fold else None, s_colsum=s)
    elif dx is not None:
        ops.linear_dx(dh1, B * H, P.W[0], s, dx, B * din, B, H, din, nb)
    if G is not None and not fold:
        ops.colsum3(dy, dout, dh2, H, dh1, H, B, G.b[2], G.b[1], G.b[0], s, nb)


class Actor(nn.Module):
    """MLP actor network (curl_sac.py:57-121)."""

    def __init__(self, obs_shape, action_shape, hidden_dim, encoder_feature_dim, log_std_min, log_std_max,
                 num_layers, num_filters):
        super().__init__()
        self.encoder = CNNEncoder(obs_shape, encoder_feature_dim, num_layers, num_filters, output_logits=True)
        self.log_std_min = log_std_min
        self.log_std_max = log_std_max
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.trunk = nn.Sequential(
            nn.Linear(self.encoder.feature_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, 2 * action_shape[0])
        )
        self.outputs = dict()
        self.apply(weight_init)

    def forward(self, obs, compute_pi=True, compute_log_pi=True, detach_encoder=False, noise=None):
        """Inference forward on the HIP kernels; same return tuple as the
        reference (mu, pi, log_pi, log_std) with pi/log_pi None when not asked
        for.  ``noise`` replaces torch.randn_like (curl_sac.py:97)."""
        z = self.encoder(obs, detach=detach_encoder)
        B, A, H, F = z.shape[0], self.action_dim, self.hidden_dim, self.encoder.feature_dim
        dev = z.device
        h1 = torch.empty((B, H), device=dev)
        h2 = torch.empty((B, H), device=dev)
        out = torch.empty((B, 2 * A), device=dev)
        mu = torch.empty((B, A), device=dev)
        log_std = torch.empty((B, A), device=dev)
        pi = log_pi = None
        if compute_pi:
            if noise is None:
                noise = torch.randn((B, A), device=dev)
            pi = torch.empty((B, A), device=dev)
            log_pi = torch.empty((B, 1), device=dev) if compute_log_pi else None
        _mlp_fwd(z, 0, _Mlp(self.trunk), 1, B, F, H, 2 * A, h1, h2, out,
                 head=(noise if compute_pi else None, self.log_std_min, self.log_std_max,
                       dict(mu=mu, pi=pi, log_pi=log_pi, log_std=log_std)))
        self.outputs['mu'] = out[:, :A]  # the pre-squash mean, as the reference records it (curl_sac.py:92)
        self.outputs['std'] = log_std.exp()
        return mu, pi, log_pi, log_std

    def log(self, L, step, log_freq=LOG_FREQ):
        """Histograms of the recorded outputs and of the three trunk layers, every ``log_freq`` steps
        (curl_sac.py:112-121; same keys)."""
        if step % log_freq:
            return
        for name, value in self.outputs.items():
            L.log_histogram('train_actor/%s_hist' % name, value, step)
        for n, layer in enumerate((self.trunk[0], self.trunk[2], self.trunk[4]), start=1):
            L.log_param('train_actor/fc%d' % n, layer, step)


class QFunction(nn.Module):
    """MLP for q-function (curl_sac.py:124-139); parameter container."""

    def __init__(self, obs_dim, action_dim, hidden_dim):
        super().__init__()
        self.trunk = nn.Sequential(
            nn.Linear(obs_dim + action_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, 1)
        )


class Critic(nn.Module):
    """Critic network, employs two Q-functions (curl_sac.py:142-180)."""

    def __init__(self, obs_shape, action_shape, hidden_dim, encoder_feature_dim, num_layers, num_filters):
        super().__init__()
        self.encoder = CNNEncoder(obs_shape, encoder_feature_dim, num_layers, num_filters, output_logits=True)
        self.Q1 = QFunction(self.encoder.feature_dim, action_shape[0], hidden_dim)
        self.Q2 = QFunction(self.encoder.feature_dim, action_shape[0], hidden_dim)
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.outputs = dict()
        self.twin_stride = None  # floats between Q1 and Q2 tensors once flattened by the agent
        self.apply(weight_init)

    def twin(self, grads=False):
        if self.twin_stride is None:
            raise RuntimeError("Critic parameters are not in the flat twin layout (constructed outside CurlSacAgent)")
        return _Mlp(self.Q1.trunk, self.twin_stride, grads)

    def forward(self, obs, action, detach_encoder=False):
        assert obs.size(0) == action.size(0)  # curl_sac.py:136
        z = self.encoder(obs, detach=detach_encoder)
        B, A, H, F = z.shape[0], self.action_dim, self.hidden_dim, self.encoder.feature_dim
        dev = z.device
        xa = torch.empty((B, F + A), device=dev)
        ops.concat(z, action.contiguous().float(), B, F, A, xa)
        h1 = torch.empty((2, B, H), device=dev)
        h2 = torch.empty((2, B, H), device=dev)
        q = torch.empty((2, B, 1), device=dev)
        _mlp_fwd(xa, 0, self.twin(), 2, B, F + A, H, 1, h1, h2, q)
        self.outputs['q1'], self.outputs['q2'] = q[0], q[1]
        return q[0], q[1]

    def log(self, L, step, log_freq=LOG_FREQ):
        """Histograms of the recorded Q values and of both Q trunks, every ``log_freq`` steps
        (curl_sac.py:171-180; same keys)."""
        if step % log_freq:
            return
        for name, value in self.outputs.items():
            L.log_histogram('train_critic/%s_hist' % name, value, step)
        for layer in range(3):
            for tag, q in (('q1', self.Q1), ('q2', self.Q2)):
                L.log_param('train_critic/%s_fc%d' % (tag, layer), q.trunk[2 * layer], step)


class CURL(nn.Module):
    """CURL head (curl_sac.py:183-222)."""

    def __init__(self, obs_shape, z_dim, critic, critic_target, output_type="continuous"):
        super().__init__()
        self.encoder = critic.encoder
        self.encoder_target = critic_target.encoder
        self.W = nn.Parameter(torch.rand(z_dim, z_dim))
        self.output_type = output_type

    def encode(self, x, detach=False, ema=False):
        return self.encoder_target(x) if ema else self.encoder(x)

    def compute_logits(self, z_a, z_pos):
        """(B,B) logits z_a (W z_pos^T) minus the row max (curl_sac.py:211-222),
        both products on the MFMA GEMM."""
        B, F = z_a.shape
        WzT = torch.empty((B, F), device=z_a.device)
        logits = torch.empty((B, B), device=z_a.device)
        ops.linear_fwd(z_pos.contiguous(), 0, self.W, 0, None, 0, WzT, 0, B, F, F)
        ops.linear_fwd(z_a.contiguous(), 0, WzT, 0, None, 0, logits, 0, B, B, F)
        return logits - torch.max(logits, 1)[0][:, None]


class _Workspace:
    """Every per-update buffer for one batch size, allocated once (static
    addresses: the kernel sequence is hipGraph-capturable)."""

    def __init__(self, agent, B):
        enc = agent.critic.encoder
        dev = agent.device
        F, A, H, L = enc.feature_dim, agent.action_dim, agent.hidden_dim, enc.num_layers
        f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
        shapes = [(B, h, w, enc.num_filters) for (h, w) in enc.layer_hw[1:]]
        # [obs | next_obs] activations of the critic phase's pass through the online convs (one launch per layer for
        # both); the obs half doubles as the activations kept for a backward pass
        self.acts_pair = [f(2 * s[0], *s[1:]) for s in shapes]
        self.acts_main = [a[:B] for a in self.acts_pair]
        self.acts_tmp = [f(*s) for s in shapes]   # no-grad passes
        gmax = max(int(np.prod(s)) for s in shapes)
        self.gbuf = [f(gmax), f(gmax)]
        self.gviews = [[g[:int(np.prod(s))].view(s) for s in shapes] for g in self.gbuf]
        # one slab workspace per conv layer: the layers' weight-gradient reductions run as ONE launch at the end of a
        # backward pass (ops.wgrad_reduce_multi)
        self.wg_ws = [f(ops.wgrad_workspace_floats(enc.obs_shape[0] if i == 0 else enc.num_filters)) for i in range(L)]
        # features
        self.z_a, self.z_t, self.z_c, self.z_pos = f(B, F), f(B, F), f(B, F), f(B, F)
        self.xhat_c, self.rstd_c, self.xhat_a, self.rstd_a = f(B, F), f(B), f(B, F), f(B)
        self.dz, self.dfc = f(B, F), f(B, F)
        self.ln_partial = f(ops.ln_partial_floats(B, F))  # LayerNorm parameter-gradient partial sums (ops.ln_bwd defer=)
        # CURL.W gradient partial sums (ops.curl_head: feature widths it is built for only)
        self.w_partial = f(max(1, B // 16) * F * F) if ops.curl_head_supported(B, F) else None
        self.fc_out = f(B, F)  # pre-LayerNorm features, only written on histogram-logging steps
        # actor trunk
        self.a_h1, self.a_h2, self.a_out = f(B, H), f(B, H), f(B, 2 * A)
        self.a_dh1, self.a_dh2, self.a_dout = f(B, H), f(B, H), f(B, 2 * A)
        self.noise = f(B, A)
        self.mu, self.pi, self.log_std, self.tanh_ls, self.gpi = f(B, A), f(B, A), f(B, A), f(B, A), f(B, A)
        self.log_pi = f(B, 1)
        # twin Q
        # [0] = the target critic's pass over next_obs, [1] = the critic's over obs: one launch per layer for the four
        # Q functions (update_critic); the names without a 2 are the critic's halves, what the backward reads
        self.xa2, self.q_h1_2, self.q_h2_2, self.q2 = f(2, B, F + A), f(2, 2, B, H), f(2, 2, B, H), f(2, 2, B, 1)
        self.xa, self.q_h1, self.q_h2 = self.xa2[1], self.q_h1_2[1], self.q_h2_2[1]
        self.tq, self.q = self.q2[0], self.q2[1]
        self.dxa = f(2, B, F + A)
        self.q_dh1, self.q_dh2 = f(2, B, H), f(2, B, H)
        self.dq, self.target_q = f(2, B, 1), f(B, 1)
        # scalars: [0] critic loss, [1..4] actor_loss/alpha_loss/entropy/alpha, [5] curl loss, [6] batch reward
        self.scalars = torch.zeros(8, device=dev, dtype=torch.float32)
        # CURL
        self.WzT, self.dWzT = f(B, F), f(B, F)
        self.logits, self.dlogits, self.row_loss = f(B, B), f(B, B), f(B)


class _NullLog:
    """What a captured update logs to: nothing (logging steps are never captured)."""

    def log(self, *a, **k):
        pass

    log_histogram = log_param = log_image = log


class CurlSacAgent(object):
    """CURL representation learning with SAC (curl_sac.py:224-465)."""

    def __init__(
        self,
        obs_shape,
        action_shape,
        device,
        augmentor,
        hidden_dim=256,
        discount=0.99,
        init_temperature=0.01,
        alpha_lr=1e-3,
        alpha_beta=0.9,
        actor_lr=1e-3,
        actor_beta=0.9,
        actor_log_std_min=-10,
        actor_log_std_max=2,
        actor_update_freq=2,
        critic_lr=1e-3,
        critic_beta=0.9,
        critic_tau=0.005,
        critic_target_update_freq=2,
        encoder_feature_dim=50,
        encoder_lr=1e-3,
        encoder_tau=0.005,
        num_layers=4,
        num_filters=32,
        cpc_update_freq=1,
        log_interval=100,
        log_param_hist_imgs=False,
        detach_encoder=False,
        pixel_sac=False
    ):
        self.augmentor = augmentor
        self.device = torch.device(device)
        self.discount = discount
        self.critic_tau = critic_tau
        self.encoder_tau = encoder_tau
        self.actor_update_freq = actor_update_freq
        self.critic_target_update_freq = critic_target_update_freq
        self.cpc_update_freq = cpc_update_freq
        self.log_interval = log_interval
        self.log_param_hist_imgs = log_param_hist_imgs
        self.image_shape = tuple(obs_shape[-2:])
        self._act_stage = {}
        self.detach_encoder = detach_encoder
        self.pixel_sac = pixel_sac
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim

        # same construction order as the reference => same RNG stream => same init for a given seed
        self.actor = Actor(obs_shape, action_shape, hidden_dim, encoder_feature_dim, actor_log_std_min,
                           actor_log_std_max, num_layers, num_filters)
        self.critic = Critic(obs_shape, action_shape, hidden_dim, encoder_feature_dim, num_layers, num_filters)
        self.critic_target = Critic(obs_shape, action_shape, hidden_dim, encoder_feature_dim, num_layers, num_filters)
        self.critic_target.load_state_dict(self.critic.state_dict())

        # tie encoders between actor and critic, and CURL and critic
        self.actor.encoder.copy_conv_weights_from(self.critic.encoder)

        self.log_alpha = torch.tensor(np.log(init_temperature)).to(self.device)  # float64, like the reference
        self.log_alpha.requires_grad = True
        self.target_entropy = -np.prod(action_shape)

        self.CURL = CURL(obs_shape, encoder_feature_dim, self.critic, self.critic_target, output_type='continuous')

        self._to_flat_device_layout()

        actor_own = [p for n, p in self.actor.named_parameters() if ".convs." not in n]
        enc_params = list(self.critic.encoder.parameters())
        # Optimizers (curl_sac.py:299-313).  The reference hands Adam the tied convs (actor) and the target
        # encoder (cpc) as well; their .grad is always None at step time there, so Adam never touches them.
        # FlatAdam is torch.optim.Adam with its step() as one HIP launch over the flat buffer (optim.py);
        # CURLA_TORCH_ADAM=1 keeps torch's own fused multi-tensor step (same state_dict either way).
        if self.device.type == "cuda" and os.environ.get("CURLA_TORCH_ADAM", "0") == "1":
            def adam(params, flat, gflat, **kw):
                return torch.optim.Adam(params, fused=True, **kw)
        elif self.device.type == "cuda":
            def adam(params, flat, gflat, **kw):
                return FlatAdam(params, flat, gflat, **kw)
        else:
            def adam(params, flat, gflat, **kw):
                return torch.optim.Adam(params, **kw)
        cf, cg = self._critic_flat, self._critic_gflat
        self.actor_optimizer = adam(actor_own, self._actor_flat, self._actor_gflat, lr=actor_lr, betas=(actor_beta, 0.999))
        self.critic_optimizer = adam(list(self.critic.parameters()), cf, cg, lr=critic_lr, betas=(critic_beta, 0.999))
        # (beside a FlatAdam actor, log_alpha's step rides in the actor's launch: FlatAdam.step_with_scalar)
        flat_actor = isinstance(self.actor_optimizer, FlatAdam)
        self.log_alpha_optimizer = torch.optim.Adam(
            [self.log_alpha], lr=alpha_lr, betas=(alpha_beta, 0.999),
            **(dict(foreach=False) if flat_actor else dict(fused=True) if self.device.type == "cuda" else {}))
        self.encoder_optimizer = adam(enc_params, cf, cg, lr=encoder_lr)
        self.cpc_optimizer = adam([self.CURL.W] + enc_params, cf, cg, lr=encoder_lr)

        self._workspaces = {}
        self._anchor_cache = None
        self._pos_cache = None   # the obs_pos handle whose target-encoder activations sit in the workspace
        self._pos_hint = None    # set by update(): the positives the actor phase may encode along the way
        self._pos_fc_done = False  # ... and whose fc product it then leaves in the target encoder's partial buffer
        self._dp_group = None
        self._dp_world = 1
        self._dp_active = False
        self._dp_avg = False
        self._dp_staged = False
        self._dp_overlap = False
        self._dp_check_every = 0
        self._dp_pending = []
        self._graphs = None      # captured update graphs: None (off) or {kind: state} (enable_update_graphs)
        self._graph_cap = None   # while a graph is being captured: the device addresses its kernels read their
        #                          per-update values from
        self.train()
        self.critic_target.train()

    # ------------------------------------------------------------------ layout
    def _to_flat_device_layout(self):
        """Move every parameter into flat fp32 device buffers (16-byte aligned
        slots), Q1/Q2 as two identically laid-out blocks, and give each a
        persistent .grad view into a mirror buffer.
          critic flat : [ CURL.W | encoder (convs, fc, ln) | Q1 | Q2 ]
          target flat : same layout (W slot unused)
          actor flat  : [ encoder.fc, encoder.ln | trunk ]"""
        dev = self.device
        for m in (self.actor, self.critic, self.critic_target):
            m.encoder.to_kernel_layout()

        def place(named, flat_size_only=False, flat=None, gflat=None, off0=0):
            off = off0
            for _, p in named:
                n = p.numel()
                if not flat_size_only:
                    dst = flat[off:off + n].view(p.shape)
                    dst.copy_(p.data)
                    p.data = dst
                    if gflat is not None:
                        p.grad = gflat[off:off + n].view(p.shape)
                off += (n + 3) & ~3
            return off

        def critic_groups(critic, W):
            enc = [("encoder." + n, p) for n, p in critic.encoder.named_parameters()]
            q1 = [("Q1." + n, p) for n, p in critic.Q1.named_parameters()]
            q2 = [("Q2." + n, p) for n, p in critic.Q2.named_parameters()]
            return [("W", W)] if W is not None else [], enc, q1, q2

        wl, enc, q1, q2 = critic_groups(self.critic, self.CURL.W)
        w_sz = place(wl, True)
        enc_sz = place(enc, True)
        q_sz = place(q1, True)
        total = w_sz + enc_sz + 2 * q_sz
        self._lay = dict(w=(0, w_sz), enc=(w_sz, w_sz + enc_sz), q=(w_sz + enc_sz, total), total=total, qblock=q_sz)

        # target and critic parameters in one allocation, [target | critic]: a fixed distance apart, so the four Q
        # functions are one two-level batch (update_critic)
        both = torch.zeros(2 * total, device=dev, dtype=torch.float32)
        self._twin_outer = total if os.environ.get("CURLA_FOUR_Q", "1") != "0" else None
        # (read per agent, like the other Python-level switches: tests/test_gpu_switches.py builds one agent per value)
        self._curl_unfused = os.environ.get("CURLA_CURL_HEAD") == "unfused"

        def build(critic, W, with_grad):
            flat = both[total:] if with_grad else both[:total]
            gflat = torch.zeros(total, device=dev, dtype=torch.float32) if with_grad else None
            wl_, enc_, q1_, q2_ = critic_groups(critic, W)
            place(wl_, False, flat, gflat, 0)
            place(enc_, False, flat, gflat, w_sz)
            place(q1_, False, flat, gflat, w_sz + enc_sz)
            place(q2_, False, flat, gflat, w_sz + enc_sz + q_sz)
            critic.twin_stride = q_sz
            return flat, gflat

        self._critic_flat, self._critic_gflat = build(self.critic, self.CURL.W, True)
        self._target_flat, _ = build(self.critic_target, None, False)
        for p in self.critic_target.parameters():
            p.requires_grad_(False)

        actor_own = [(n, p) for n, p in self.actor.named_parameters() if ".convs." not in n]
        a_sz = place(actor_own, True)
        self._actor_flat = torch.zeros(a_sz, device=dev, dtype=torch.float32)
        # the actor's data-parallel bucket: [fc, ln | trunk | ops.F64_WORDS words]; the words carry log_alpha's float64
        # gradient through the bucket's all-reduce (ops.f64_pack) -- unused without data parallelism
        self._actor_gbucket = torch.zeros(a_sz + ops.F64_WORDS, device=dev, dtype=torch.float32)
        self._actor_gflat = self._actor_gbucket[:a_sz]
        self._la_words = self._actor_gbucket[a_sz:]
        place(actor_own, False, self._actor_flat, self._actor_gflat, 0)
        self.log_alpha.grad = torch.zeros((), device=dev, dtype=torch.float64)

    _curl_unfused = False       # the CURL head as separate launches (set per agent from CURLA_CURL_HEAD=unfused)
    _soft_update_hint = False   # set by update() around update_critic(): a target soft update follows the critic's step
    _soft_update_done = False

    def _ws(self, B):
        if B not in self._workspaces:
            from . import _lib
            if self.device.type != "cuda" and _lib._trace_hook is None:
                raise RuntimeError("CurlSacAgent.update needs a CUDA/HIP device: the learner path has no CPU fallback")
            self._workspaces[B] = _Workspace(self, B)
        return self._workspaces[B]

    # --------------------------------------------------------------- data parallel
    def enable_data_parallel(self, process_group=None, single_rank_collectives=False, overlap=None, broadcast=True,
                             check_every=None):
        """Synchronous data parallelism (one process per GPU): before every
        optimizer step the freshly written flat gradient bucket is averaged over
        ranks (RCCL over xGMI).  RCCL averages inside the collective
        (ncclAvg); other backends (gloo in the CPU tests) sum and divide.

        ``broadcast``: rank 0's parameters, targets and log_alpha replace every
        other rank's (three flat buffers + one scalar), so ranks need not have
        been seeded identically.  ``check_every`` (default 1000 updates, env
        CURLA_DP_CHECK_EVERY, 0 = never): a checksum of the replicated state is
        compared across ranks and a mismatch raises -- replicas that drift apart
        would otherwise train on silently.
        ``overlap`` (default off, env CURLA_DP_OVERLAP=1 turns it on): each
        bucket is reduced in two asynchronous pieces issued where their
        gradients become final -- the twin-Q / fc / LayerNorm gradients (almost
        all of the bytes) before the conv backward starts, the conv gradients
        after it -- on the communicator's own stream; the compute stream only
        waits for them right before ``optimizer.step()``.  Off: one blocking
        all-reduce per bucket.  Both orders reduce the same elements with the
        same collective, so the result does not depend on the flag.  The default
        follows the only measurement available so far (one GPU, one-rank RCCL
        group): the asynchronous schedule costs 0.12 ms per update more than the
        blocking one, like everything else that runs next to the persistent conv
        grids (DESIGN.md sections 6, 7); flip it once an N-GPU run says otherwise.
        ``single_rank_collectives`` issues the collectives even in a world of one
        (a 1-GPU check of the exact calls an N-GPU run makes)."""
        import torch.distributed as dist
        self._dp_group = process_group if process_group is not None else dist.group.WORLD
        self._dp_world = dist.get_world_size(self._dp_group)
        self._dp_active = self._dp_world > 1 or single_rank_collectives
        self._dp_avg = dist.get_backend(self._dp_group) == "nccl"
        # A backend without a device path of its own (gloo: the CPU tests, and the two-ranks-on-one-GPU test that RCCL
        # refuses) gets the buckets through the host: device -> host copy, collective, copy back, all synchronous.
        self._dp_staged = not self._dp_avg and self.device.type == "cuda"
        if overlap is None:
            overlap = os.environ.get("CURLA_DP_OVERLAP", "0") == "1"
        self._dp_overlap = bool(overlap)
        if check_every is None:
            check_every = int(os.environ.get("CURLA_DP_CHECK_EVERY", "1000"))
        self._dp_check_every = int(check_every)
        self._dp_pending = []
        if self._dp_active and broadcast:
            src = dist.get_global_rank(self._dp_group, 0)
            with torch.no_grad():
                for t in (self._critic_flat, self._target_flat, self._actor_flat, self.log_alpha):
                    if self._dp_staged:
                        h = t.detach().cpu()
                        dist.broadcast(h, src=src, group=self._dp_group)
                        t.copy_(h)
                    else:
                        dist.broadcast(t, src=src, group=self._dp_group)

    def _replica_checksum(self):
        """float64 sums of the replicated state (deterministic: same kernel, same data => same bits)."""
        with torch.no_grad():
            return torch.stack([self._critic_flat.sum(dtype=torch.float64), self._target_flat.sum(dtype=torch.float64),
                                self._actor_flat.sum(dtype=torch.float64), self.log_alpha.detach().to(torch.float64)])

    def check_replicas(self):
        """Raise if the ranks' parameters differ (costs one 64-byte all-reduce and a host sync)."""
        if not self._dp_active:
            return
        import torch.distributed as dist
        s = self._replica_checksum()
        both = torch.stack([s, -s])  # max over ranks of (s, -s) = (max, -min)
        if self._dp_staged:
            both = both.cpu()
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self._dp_group)
        hi, lo = both[0], -both[1]
        if not bool(torch.equal(hi, lo)):
            names = ("critic", "critic_target", "actor", "log_alpha")
            bad = [n for n, a, b in zip(names, hi.tolist(), lo.tolist()) if a != b]
            raise RuntimeError("data-parallel replicas have diverged (%s differ across ranks): seed every rank "
                               "identically or keep broadcast=True in enable_data_parallel" % ", ".join(bad))

    def _allreduce(self, *buckets, async_op=False, f64_rider=None):
        """Average each tensor over the ranks.  async_op: the collectives are only enqueued (they start once
        the compute stream reaches this point); _allreduce_wait() makes the compute stream wait for them.
        ``f64_rider`` = (scalar, words): ``scalar`` (one float64 element: log_alpha's gradient) rides in ``words``, the
        last ops.F64_WORDS elements of the LAST bucket -- packed before the collective, unpacked (= the mean over the
        ranks, from the exact sum: curla_hip.h curla_f64_pack) once it has completed."""
        if not self._dp_active:
            return
        import torch.distributed as dist
        if f64_rider is not None:
            scalar, words = f64_rider
            assert words.data_ptr() + 4 * words.numel() == buckets[-1].data_ptr() + 4 * buckets[-1].numel()
            ops.f64_pack(scalar, words)
        last = len(buckets) - 1
        for i, t in enumerate(buckets):
            if t.numel() == 0:
                continue
            avg = self._dp_avg and t.dtype == torch.float32  # (a float64 bucket of its own would take the sum path)
            op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
            post = []
            if not avg and not self._dp_staged:
                post.append(lambda t=t: t.div_(self._dp_world))
            if f64_rider is not None and i == last:
                # every path leaves the AVERAGED digits in the words: n_mul makes them the digit sums again
                post.append(lambda: ops.f64_unpack(f64_rider[1], self._dp_world, self._dp_world, f64_rider[0]))
            if self._dp_staged:  # (synchronous whatever async_op says: same elements, same sum)
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self._dp_group)
                t.copy_(h.div_(self._dp_world))
                work = None
            elif async_op:
                work = dist.all_reduce(t, op=op, group=self._dp_group, async_op=True)
            else:
                dist.all_reduce(t, op=op, group=self._dp_group)
                work = None
            if work is not None:
                self._dp_pending.append((work, post))
            else:
                for fn in post:
                    fn()

    def _allreduce_wait(self):
        for work, post in self._dp_pending:
            work.wait()
            for fn in post:
                fn()
        self._dp_pending = []

    def _grad_offset(self, p, flat):
        return (p.grad.data_ptr() - flat.data_ptr()) // 4

    # ------------------------------------------------------------------ reference API
    def train(self, training=True):
        self.training = training
        self.actor.train(training)
        self.critic.train(training)
        self.CURL.train(training)

    @property
    def alpha(self):
        return self.log_alpha.exp()

    def _stage_obs(self, obs):
        """One observation for the actor.  uint8 (C, H, W) frames -- what the environment and train.py hand over --
        go through a pinned host buffer into a one-slot uint8 NHWC device ring and are read by the same fused
        conv1 loader as a replay minibatch (52 KB over PCIe instead of a pageable 208 KB float copy); anything else
        takes the reference's torch.FloatTensor(obs) route (curl_sac.py:332-333)."""
        if not (isinstance(obs, np.ndarray) and obs.dtype == np.uint8 and obs.ndim == 3):
            return torch.FloatTensor(np.ascontiguousarray(obs)).to(self.device).unsqueeze(0)
        C, H, W = obs.shape
        st = self._act_stage.get((C, H, W))
        if st is None:
            n = H * W * C
            pin = torch.empty(n, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
            ring = torch.zeros(n + 32, dtype=torch.uint8, device=self.device)  # + loader slack (curla_hip.h)
            st = self._act_stage[(C, H, W)] = (pin, pin.numpy().reshape(H, W, C), ring, ring[:n].view(1, H, W, C))
        pin, pin_hwc, ring, frames = st
        np.copyto(pin_hwc, obs.transpose(1, 2, 0))
        ring[:pin.numel()].copy_(pin, non_blocking=True)
        return ops.ObsRef.from_ring(frames, None, None, None, 1, (H, W))

    def select_action(self, obs):
        """curl_sac.py:330-337."""
        with torch.no_grad():
            mu, _, _, _ = self.actor(self._stage_obs(obs), compute_pi=False, compute_log_pi=False)
            return mu.cpu().data.numpy().flatten()

    def sample_action(self, obs, noise=None):
        """curl_sac.py:339-347."""
        if obs.shape[-2:] != self.image_shape:
            obs = self.augmentor.evaluation_augmentation(obs)
        with torch.no_grad():
            mu, pi, _, _ = self.actor(self._stage_obs(obs), compute_log_pi=False, noise=noise)
            return pi.cpu().data.numpy().flatten()

    # ------------------------------------------------------------------ building blocks
    def _encoder_backward(self, ws, obs_ref, dz, xhat, rstd, enc, conv_grads=True, dense_done=None, twin_ld=None,
                          ln_token=None):
        """Backward of fc+LN and (optionally) the conv stack from d(loss)/d(z);
        writes .grad of enc.{ln,fc,convs}.  ``dense_done()`` is called once the fc / LayerNorm gradients are
        final, i.e. before the conv backward is enqueued (data parallel: their all-reduce starts there).
        ``twin_ld``: ``dz`` is the twin-Q input gradient [2, B, twin_ld]; d(loss)/d(z) is the sum over the twin of
        its first F columns, read in place by the LayerNorm backward."""
        B, F, K, L = obs_ref.B, enc.feature_dim, enc.flat_dim, enc.num_layers
        acts = ws.acts_main
        dy, dy2 = (dz, None) if twin_ld is None else (dz[0], dz[1])
        streams = ops.fc_bwd_streams(F, K)
        # (the LayerNorm's parameter gradients and the fc bias gradient -- column sums over the batch -- are left as
        # partial sums and finished inside the fc backward's launch, when that is one of the streaming kernels)
        # (``ln_token``: the CURL head's launch has already done the LayerNorm backward into ws.dfc and left its
        # partial sums -- and those of CURL.W's gradient -- for the fc backward to finish)
        ln = ln_token if ln_token is not None else ops.ln_bwd(
            dy, xhat, rstd, enc.ln.weight, B, F, ws.dfc, dgamma=enc.ln.weight.grad, dbeta=enc.ln.bias.grad,
            dbias_in=enc.fc.bias.grad, dy2=dy2, ld=twin_ld, defer=ws.ln_partial if streams else None)
        h = acts[-1]
        cur = L % 2
        g = ws.gviews[cur][L - 1]
        if streams and conv_grads:
            # fc weight gradient and the data gradient into the conv stack (ReLU mask of the last conv layer fused)
            # in one launch: both only read dfc
            ops.fc_bwd(ws.dfc, enc.fc.weight, h, g, enc.fc.weight.grad, B, F, K, ln=ln)
        elif streams:
            ops.fc_dw(ws.dfc, h, enc.fc.weight.grad, B, F, K, ln=ln)
        else:
            ops.linear_dw(ws.dfc, 0, h, 0, enc.fc.weight.grad, 0, B, F, K)
        if dense_done is not None:
            dense_done()
        if not conv_grads:
            return
        if not streams:
            ops.linear_dx(ws.dfc, 0, enc.fc.weight, 0, g, 0, B, F, K, mask=h)
        jobs = []
        for layer in range(L, 1, -1):  # layer l: input acts[l-2], output acts[l-1]
            conv = enc.convs[layer - 1]
            cur ^= 1
            gin = ws.gviews[cur][layer - 2]
            # weight gradient (slabs) and data gradient of the layer: both only read g, one launch for both
            n = ops.conv_s1_bwd_slabs(acts[layer - 2], g, conv.weight, gin, ws.wg_ws[layer - 1])
            jobs.append((ws.wg_ws[layer - 1], n, conv.weight.grad, conv.bias.grad))
            g = gin
        n = ops.conv1_wgrad_slabs(obs_ref, g, ws.wg_ws[0], enc.num_filters)
        jobs.append((ws.wg_ws[0], n, enc.convs[0].weight.grad, enc.convs[0].bias.grad))
        for i in range(0, len(jobs), 8):
            ops.wgrad_reduce_multi(jobs[i:i + 8])

    def _records(self, step):
        """True on the steps whose module outputs the optional histogram / image logging reads
        (log_param_hist_imgs, every LOG_FREQ steps: curl_sac.py:17,112-121,171-180)."""
        return self.log_param_hist_imgs and step % LOG_FREQ == 0

    def _noise(self, ws, noise):
        """(noise buffer, rng): explicit ``noise`` is copied into the buffer (rng None).  Without it
        (``torch.randn_like``, curl_sac.py:97) the policy-head launch draws the numbers itself and writes them into the
        buffer: rng = (seed, Philox counter) taken from the device's default torch generator, whose offset is moved on
        here by the numbers drawn (rounded up to 4) -- consecutive draws use NON-OVERLAPPING counters and
        torch.manual_seed, get_rng_state / set_rng_state (checkpoints) and the per-rank seeds keep their meaning.  The
        offset trajectory is this build's own: it is NOT the one a ``normal_()`` launch would leave behind (torch's
        increment depends on its grid), so a generator state saved by the ``normal_()`` path continues with different
        numbers.  Where the generator does not expose its offset this agent warns once and draws with ``normal_()``."""
        if noise is not None:
            ws.noise.copy_(noise)
            return ws.noise, None
        if self._graph_cap is not None:
            # captured update graph: (seed, counter) are read by the kernel from the update's control block; the host
            # side of the draw (the generator's offset) is advanced by the graph manager once per replay
            cap = self._graph_cap
            addr = cap["rng"][cap["n_noise"]]
            cap["n_noise"] += 1
            return ws.noise, (0, 0, addr)
        if ws.noise.is_cuda and not self._noise_launch:
            try:
                gen = torch.cuda.default_generators[ws.noise.device.index if ws.noise.device.index is not None
                                                    else torch.cuda.current_device()]
                off = gen.get_offset()
                gen.set_offset(off + 4 * ((ws.noise.numel() + 3) // 4))
                return ws.noise, (gen.initial_seed(), off // 4)
            except (AttributeError, RuntimeError) as e:
                self._noise_launch = True  # (this agent only)
                warnings.warn(f"curla_amd: the torch generator exposes no Philox offset ({e!r}); policy noise falls "
                              "back to a separate normal_() launch per phase", RuntimeWarning, stacklevel=2)
        ws.noise.normal_()
        return ws.noise, None

    _noise_launch = False  # True: this torch build's generator has no get_offset / set_offset

    # ------------------------------------------------------------------ phases
    def update_critic(self, obs, action, reward, next_obs, not_done, L, step, noise=None):
        """curl_sac.py:349-371."""
        o, no = _as_ref(obs), _as_ref(next_obs)
        B, A, H = o.B, self.action_dim, self.hidden_dim
        enc, F = self.critic.encoder, self.critic.encoder.feature_dim
        ws = self._ws(B)
        self._anchor_cache = self._pos_cache = None
        self._pos_fc_done = False
        action, reward, not_done = action.contiguous(), reward.contiguous(), not_done.contiguous()

        # -- target (no_grad block, curl_sac.py:350-355)
        # obs and next_obs both go through the online convs (the actor's are tied to the critic's): when the buffer
        # handed them over as one [obs | next_obs] handle they take ONE launch per layer (2B samples)
        pair = getattr(o, "pair", None)
        merged = pair is not None and pair[1] is no and pair[0].B == 2 * B
        tenc = self.critic_target.encoder
        if merged:
            # ... and the target encoder's pass over next_obs rides in the same launches (second problem, own weights)
            enc.conv_forward2(pair[0], ws.acts_pair, tenc, no, ws.acts_tmp)
            h_next = ws.acts_pair[-1][B:]
        else:
            enc.conv_forward(no, ws.acts_tmp)                   # tied convs, online weights
            h_next = ws.acts_tmp[-1]
        if merged:  # all three activation tensors are there: the three fc products in one launch
            CNNEncoder.fc_partial_multi([(self.actor.encoder, h_next), (tenc, ws.acts_tmp[-1]), (enc, ws.acts_main[-1])])
        else:
            self.actor.encoder.fc_partial(h_next)
        # the target critic's twins over [z', a'] and the critic's over [z, a] (curl_sac.py:353-358) are four MLPs of
        # one shape: with all conv outputs there they share their three launches, and the three encoders' features
        # (actor, target, critic) their one -- the action columns of [z' | a'] are then written by the policy head
        four = merged and self._twin_outer is not None
        rec = self._records(step)
        if four:
            CNNEncoder.ln_from_partial_multi(B, [
                (self.actor.encoder, ws.z_a, {}), (tenc, ws.z_t, dict(xa=ws.xa2[0])),
                (enc, ws.z_c, dict(xhat=ws.xhat_c, rstd=ws.rstd_c, fc_out=ws.fc_out if rec else None, xa=ws.xa,
                                   act=action))], A)
        else:
            self.actor.encoder.ln_from_partial(B, ws.z_a)
        nz, rng = self._noise(ws, noise)
        _mlp_fwd(ws.z_a, 0, _Mlp(self.actor.trunk), 1, B, F, H, 2 * A, ws.a_h1, ws.a_h2, ws.a_out,
                 head=(nz, self.actor.log_std_min, self.actor.log_std_max,
                       dict(pi=None if four else ws.pi, log_pi=ws.log_pi, xa=ws.xa2[0] if four else None, rng=rng)))
        if four:
            _mlp_fwd(ws.xa2, 0, self.critic_target.twin(), 2, B, F + A, H, 1, ws.q_h1_2, ws.q_h2_2, ws.q2,
                     outer=(2, self._twin_outer))
        else:
            if not merged:
                tenc.conv_forward(no, ws.acts_tmp)
                tenc.fc_partial(ws.acts_tmp[-1])
            tenc.ln_from_partial(B, ws.z_t, xa=ws.xa, act=ws.pi)  # xa = cat([z, a'], 1)
            _mlp_fwd(ws.xa, 0, self.critic_target.twin(), 2, B, F + A, H, 1, ws.q_h1, ws.q_h2, ws.tq)
            # -- current Q estimates + loss + backward (curl_sac.py:357-367)
            if not merged:
                enc.conv_forward(o, ws.acts_main)
                enc.fc_partial(ws.acts_main[-1])
            enc.ln_from_partial(B, ws.z_c, xhat=ws.xhat_c, rstd=ws.rstd_c, fc_out=ws.fc_out if rec else None,
                                xa=ws.xa, act=action)
            _mlp_fwd(ws.xa, 0, self.critic.twin(), 2, B, F + A, H, 1, ws.q_h1, ws.q_h2, ws.q)
        if rec:  # what critic.log() / encoder.log() histogram: the outputs of THIS forward (curl_sac.py:163-167)
            self.critic.outputs['q1'], self.critic.outputs['q2'] = ws.q[0].clone(), ws.q[1].clone()
            enc.record_from(o, ws.acts_main, ws.fc_out, ws.z_c)
        # target_Q = r + not_done * gamma * (min Q' - alpha log pi'), the two MSE terms and their gradient dq: computed
        # inside the backward launch of the Q functions' last layer (a launch of its own otherwise)
        td = (dict(kind=1, twin_stride=B, q=ws.q, target_q_twin=ws.tq, log_pi=ws.log_pi, reward=reward,
                   not_done=not_done, log_alpha=self.log_alpha, discount=self.discount, target_q=ws.target_q,
                   scalars=ws.scalars[0:1], dq=ws.dq),
              lambda: ops.critic_td_loss(ws.q, ws.tq, B, ws.log_pi, reward, not_done, self.log_alpha, self.discount, B,
                                         ws.target_q, ws.scalars[0:1], ws.dq))
        _mlp_bwd(ws.xa, 0, self.critic.twin(), self.critic.twin(grads=True), 2, B, F + A, H, 1, ws.q_h1, ws.q_h2, ws.dq,
                 ws.q_dh2, ws.q_dh1, ws.dxa, loss=td)
        if step % self.log_interval == 0:
            L.log('train_critic/loss', ws.scalars[0], step)
        # (d(loss)/d(z) = dxa[0][:, :F] + dxa[1][:, :F], torch.cat's backward: summed inside the LayerNorm backward)
        # data parallel: the bucket is [convs | fc, ln | Q1 | Q2]; everything behind the convs is final before the
        # conv backward starts and is reduced underneath it
        lay = self._lay
        e0, total = lay["enc"][0], lay["total"]
        if self._dp_active and self._dp_overlap:
            cut = self._grad_offset(enc.fc.weight, self._critic_gflat)
            self._encoder_backward(ws, o, ws.dxa, ws.xhat_c, ws.rstd_c, enc, conv_grads=not self.detach_encoder,
                                   dense_done=lambda: self._allreduce(self._critic_gflat[cut:total], async_op=True),
                                   twin_ld=F + A)
            self._allreduce(self._critic_gflat[e0:cut], async_op=True)
            self._allreduce_wait()
        else:
            self._encoder_backward(ws, o, ws.dxa, ws.xhat_c, ws.rstd_c, enc, conv_grads=not self.detach_encoder,
                                   twin_ld=F + A)
            self._allreduce(self._critic_gflat[e0:total])
        if self.detach_encoder:  # convs received no gradient: Adam must skip them (grad None in the reference)
            saved = [(p, p.grad) for m in enc.convs for p in (m.weight, m.bias)]
            for p, _ in saved:
                p.grad = None
            self.critic_optimizer.step()
            for p, g in saved:
                p.grad = g
        else:
            # update() announces a target soft update right behind this step (curl_sac.py:442-445): it reads what the
            # step writes, so both share one pass over the critic's flat parameters when the optimizer is the flat one
            fused = False
            if self._soft_update_hint:
                (e0, e1), (_, q1) = lay["enc"], lay["q"]
                fused = FlatAdam.step_with_lerp(self.critic_optimizer, self._target_flat, e0, q1, e1 - e0,
                                                self.encoder_tau, self.critic_tau)
                self._soft_update_done = fused
            if not fused:
                self.critic_optimizer.step()
        if self.log_param_hist_imgs:
            self.critic.log(L, step)

    def update_actor_and_alpha(self, obs, L, step, noise=None):
        """curl_sac.py:373-404.  Only the live gradients are produced: the
        actor's own fc/ln/trunk and log_alpha (the reference also deposits
        gradients on critic tensors that are zeroed before any step reads them)."""
        o = _as_ref(obs)
        B, A, H = o.B, self.action_dim, self.hidden_dim
        enc, aenc, F = self.critic.encoder, self.actor.encoder, self.critic.encoder.feature_dim
        ws = self._ws(B)

        # one conv pass feeds actor.fc, critic.fc and (even steps) the CURL anchor branch; when update() knows that the
        # CURL phase follows with unchanged target weights, the target encoder's pass over the positives rides in the
        # same launches (second problem) and update_cpc finds its activations ready
        pos = self._pos_hint
        self._pos_hint = None
        if pos is not None:
            enc.conv_forward2(o, ws.acts_main, self.critic_target.encoder, pos, ws.acts_tmp)
            self._pos_cache = pos
        else:
            enc.conv_forward(o, ws.acts_main)
        h = ws.acts_main[-1]
        fcs = [(aenc, h), (enc, h)]  # actor.fc and critic.fc read the same conv output
        if pos is not None:  # ... and the positives' target features are ready too: third problem of the launch
            fcs.append((self.critic_target.encoder, ws.acts_tmp[-1]))
            self._pos_fc_done = True
        CNNEncoder.fc_partial_multi(fcs)
        # ... and so are the features: actor's, critic's (kept for the CURL anchor branch; the action columns of
        # xa = cat([z, pi], 1) come from the policy head below) and the positives'
        lns = [(aenc, ws.z_a, dict(xhat=ws.xhat_a, rstd=ws.rstd_a)),
               (enc, ws.z_c, dict(xhat=ws.xhat_c, rstd=ws.rstd_c, xa=ws.xa))]
        if pos is not None:
            lns.append((self.critic_target.encoder, ws.z_pos, {}))
        CNNEncoder.ln_from_partial_multi(B, lns, A)
        self._anchor_cache = obs

        trunk = _Mlp(self.actor.trunk)
        nz, rng = self._noise(ws, noise)
        lo, hi = self.actor.log_std_min, self.actor.log_std_max
        _mlp_fwd(ws.z_a, 0, trunk, 1, B, F, H, 2 * A, ws.a_h1, ws.a_h2, ws.a_out,
                 head=(nz, lo, hi, dict(mu=ws.mu, pi=ws.pi, log_pi=ws.log_pi, log_std=ws.log_std, tanh_ls=ws.tanh_ls,
                                        xa=ws.xa, rng=rng)))
        if self._records(step):  # what actor.log() histograms (curl_sac.py:92-93): pre-squash mean and std
            self.actor.outputs['mu'], self.actor.outputs['std'] = ws.a_out[:, :A].clone(), ws.log_std.exp()
        _mlp_fwd(ws.xa, 0, self.critic.twin(), 2, B, F + A, H, 1, ws.q_h1, ws.q_h2, ws.q)
        # actor / alpha losses and d(loss)/dQ: inside the backward launch of the Q functions' last layer
        # backward: Q -> pi -> trunk -> LN -> fc (encoder detached, curl_sac.py:375-376)
        al = (dict(kind=2, A=A, twin_stride=B, q=ws.q, log_pi=ws.log_pi, log_std=ws.log_std, log_alpha=self.log_alpha,
                   target_entropy=float(self.target_entropy), scalars=ws.scalars[1:5], dq=ws.dq,
                   dlog_alpha=self.log_alpha.grad),
              lambda: ops.actor_loss(ws.q, B, ws.log_pi, ws.log_std, A, self.log_alpha, float(self.target_entropy), B,
                                     ws.scalars[1:5], ws.dq, self.log_alpha.grad))
        _mlp_bwd(ws.xa, 0, self.critic.twin(), None, 2, B, F + A, H, 1, ws.q_h1, ws.q_h2, ws.dq, ws.q_dh2, ws.q_dh1,
                 ws.dxa, loss=al)
        if step % self.log_interval == 0:
            L.log('train_actor/loss', ws.scalars[1], step)
            L.log('train_actor/target_entropy', self.target_entropy, step)
            L.log('train_actor/entropy', ws.scalars[3], step)
        # d(loss)/d(pi) = the action columns of dxa summed over the twin, read in place
        ops.actor_head_bwd(None, self.log_alpha, 1.0 / B, nz, ws.pi, ws.log_std, ws.tanh_ls, B, A, lo, hi, ws.a_dout,
                           twin_dxa=ws.dxa, F=F)
        _mlp_bwd(ws.z_a, 0, trunk, _Mlp(self.actor.trunk, grads=True), 1, B, F, H, 2 * A, ws.a_h1, ws.a_h2, ws.a_dout,
                 ws.a_dh2, ws.a_dh1, ws.dz)
        streams = ops.fc_bwd_streams(F, enc.flat_dim)
        ln = ops.ln_bwd(ws.dz, ws.xhat_a, ws.rstd_a, aenc.ln.weight, B, F, ws.dfc, dgamma=aenc.ln.weight.grad,
                        dbeta=aenc.ln.bias.grad, dbias_in=aenc.fc.bias.grad, defer=ws.ln_partial if streams else None)
        if streams:
            ops.fc_dw(ws.dfc, h, aenc.fc.weight.grad, B, F, enc.flat_dim, ln=ln)
        else:
            ops.linear_dw(ws.dfc, 0, h, 0, aenc.fc.weight.grad, 0, B, F, enc.flat_dim)

        # data parallel: ONE collective for the phase -- the bucket [fc, ln | trunk | words], log_alpha's float64
        # gradient riding in the words (round 5: an 8-byte all-reduce of its own).  Overlapped schedule: the collective is
        # asynchronous and, when update() has announced that the CURL phase follows, this phase's two optimizer steps
        # are taken only after that phase has been enqueued (_finish_actor_step): the CURL phase reads nothing they
        # write (the actor's own fc / ln / trunk and log_alpha), so the numbers are the reference order's
        # (curl_sac.py:393-404 before :418-423) and the all-reduce runs underneath the CURL phase's convolutions.
        rider = (self.log_alpha.grad, self._la_words)
        if self._dp_active and self._dp_overlap:
            self._allreduce(self._actor_gbucket, async_op=True, f64_rider=rider)
            if self._defer_actor_step and not self.log_param_hist_imgs:
                self._actor_step_pending = True
            else:
                self._allreduce_wait()
        else:
            self._allreduce(self._actor_gbucket, f64_rider=rider)
        if not self._actor_step_pending:
            self._actor_steps()
        if self.log_param_hist_imgs:
            self.actor.log(L, step)
        if step % self.log_interval == 0:
            L.log('train_alpha/loss', ws.scalars[2], step)
            L.log('train_alpha/value', ws.scalars[4], step)

    _defer_actor_step = False    # set by update(): the CURL phase follows the actor phase (overlapped data parallel)
    _actor_step_pending = False  # the actor phase has left its optimizer steps to _finish_actor_step()

    def _actor_steps(self):
        # actor_optimizer.step() ... log_alpha_optimizer.step() (curl_sac.py:393-404): nothing in between reads
        # log_alpha (the logged alpha is the loss kernel's), so the two steps share a launch
        if isinstance(self.actor_optimizer, FlatAdam):
            FlatAdam.step_with_scalar(self.actor_optimizer, self.log_alpha_optimizer)
        else:
            self.actor_optimizer.step()
            self.log_alpha_optimizer.step()

    def _finish_actor_step(self):
        if self._actor_step_pending:
            self._actor_step_pending = False
            self._allreduce_wait()
            self._actor_steps()

    def update_cpc(self, obs_anchor, obs_pos, cpc_kwargs, L, step):
        """curl_sac.py:406-423."""
        oa, op_ = _as_ref(obs_anchor), _as_ref(obs_pos)
        B = oa.B
        enc, tenc, F = self.critic.encoder, self.critic_target.encoder, self.critic.encoder.feature_dim
        ws = self._ws(B)
        need_anchor = self._anchor_cache is None or self._anchor_cache is not obs_anchor
        need_pos = self._pos_cache is None or self._pos_cache is not obs_pos
        pos_fc_done = self._pos_fc_done and not need_pos
        self._anchor_cache = self._pos_cache = None
        self._pos_fc_done = False
        if need_anchor and need_pos:   # (odd steps) both passes in one launch per layer, own weights each
            enc.conv_forward2(oa, ws.acts_main, tenc, op_, ws.acts_tmp)
        elif need_anchor:
            enc.conv_forward(oa, ws.acts_main)
        elif need_pos:
            tenc.conv_forward(op_, ws.acts_tmp)
        if need_anchor:
            CNNEncoder.fc_partial_multi([(enc, ws.acts_main[-1]), (tenc, ws.acts_tmp[-1])])
            CNNEncoder.ln_from_partial_multi(B, [(enc, ws.z_c, dict(xhat=ws.xhat_c, rstd=ws.rstd_c)),
                                                 (tenc, ws.z_pos, {})])
        elif not pos_fc_done:  # (done: the actor phase left the positives' features in ws.z_pos)
            tenc.fc_partial(ws.acts_tmp[-1])
            tenc.ln_from_partial(B, ws.z_pos)

        W = self.CURL.W
        ops.linear_fwd(ws.z_pos, 0, W, 0, None, 0, ws.WzT, 0, B, F, F)         # (W z_pos^T)^T
        logged = step % self.log_interval == 0  # the mean of the row losses is only needed when it is logged
        # logits, cross-entropy, d z_a, the LayerNorm backward and the partial sums of dW in ONE launch where the
        # shapes allow (and the fc backward is a streaming kernel, whose extra workgroup finishes the partial sums)
        token = None
        if ops.curl_head_supported(B, F) and ops.fc_bwd_streams(F, enc.flat_dim) and not self._curl_unfused:
            token = ops.curl_head(ws.z_c, ws.z_pos, ws.WzT, ws.xhat_c, ws.rstd_c, enc.ln.weight, B, F, ws.row_loss, ws.dfc,
                                  ws.ln_partial, ws.w_partial, enc.ln.weight.grad, enc.ln.bias.grad, enc.fc.bias.grad,
                                  W.grad, loss=ws.scalars[5:6] if logged else None, logits=ws.logits, dz=ws.dz)
        else:
            ops.linear_fwd(ws.z_c, 0, ws.WzT, 0, None, 0, ws.logits, 0, B, B, F)   # z_a (W z_pos^T)
            ops.curl_ce(ws.logits, B, B, ws.row_loss, ws.scalars[5:6] if logged else None, ws.dlogits)
            ops.linear_dx(ws.dlogits, 0, ws.WzT, 0, ws.dz, 0, B, B, F)             # d z_a
            ops.linear_dw(ws.dlogits, 0, ws.z_c, 0, ws.dWzT, 0, B, B, F)           # d (W z_pos^T)^T
            ops.linear_dw(ws.dWzT, 0, ws.z_pos, 0, W.grad, 0, B, F, F)             # d W
        # data parallel: the bucket is [W | convs | fc, ln]; the encoder gradients are reduced ONCE and consumed by
        # both encoder_optimizer and cpc_optimizer
        e1 = self._lay["enc"][1]
        if self._dp_active and self._dp_overlap:
            cut = self._grad_offset(enc.fc.weight, self._critic_gflat)
            self._encoder_backward(ws, oa, ws.dz, ws.xhat_c, ws.rstd_c, enc, ln_token=token,
                                   dense_done=lambda: self._allreduce(self._critic_gflat[cut:e1], async_op=True))
            self._allreduce(self._critic_gflat[0:cut], async_op=True)
            self._allreduce_wait()
        else:
            self._encoder_backward(ws, oa, ws.dz, ws.xhat_c, ws.rstd_c, enc, ln_token=token)
            self._allreduce(self._critic_gflat[0:e1])
        if isinstance(self.encoder_optimizer, FlatAdam):
            FlatAdam.step_pair(self.encoder_optimizer, self.cpc_optimizer)  # both steps in one pass over the encoder
        else:
            self.encoder_optimizer.step()
            self.cpc_optimizer.step()
        if step % self.log_interval == 0:
            L.log('train/curl_loss', ws.scalars[5], step)

    def soft_update_targets(self):
        """utils.soft_update_params x3 (curl_sac.py:442-445) as one flat lerp with two rates."""
        lay = self._lay
        (e0, e1), (q0, q1) = lay["enc"], lay["q"]
        assert e1 == q0  # [encoder | Q1 | Q2] are adjacent: one launch, two rates
        ops.soft_update2(self._critic_flat[e0:q1], self._target_flat[e0:q1], e1 - e0, self.encoder_tau, self.critic_tau)

    def update(self, replay_buffer, L, step, only_cpc=False, noise=None):
        """curl_sac.py:426-451.  A curla_amd ReplayBuffer hands over references
        into its HBM ring (gather + crop fused into the first conv); any other
        buffer is used through the reference's ``sample_cpc()`` tensors.
        ``noise`` (parity tests): (critic-phase, actor-phase) tensors in place of the two ``torch.randn_like`` draws
        (curl_sac.py:97 via :352 and :375)."""
        if noise is None and self._graphs is not None and self._graph_usable(replay_buffer, step, only_cpc):
            return self._update_graphed(replay_buffer, L, step)
        self._update_eager(replay_buffer, L, step, only_cpc, noise)

    def _update_eager(self, replay_buffer, L, step, only_cpc=False, noise=None):
        if hasattr(replay_buffer, "sample_cpc_refs"):
            sample = replay_buffer.sample_cpc_refs()
        else:
            sample = replay_buffer.sample_cpc()

        if self._dp_active and self._dp_check_every > 0 and step % self._dp_check_every == 0:
            self.check_replicas()
        self._update_phases(sample, L, step, only_cpc, noise)

    def _update_phases(self, sample, L, step, only_cpc=False, noise=None):
        """The phases of one update on a drawn minibatch (curl_sac.py:431-451)."""
        obs, action, reward, next_obs, not_done, cpc_kwargs = sample
        noise_c, noise_a = noise if noise is not None else (None, None)
        if step % self.log_interval == 0:
            ws = self._ws(action.shape[0])
            ops.mean(reward.contiguous(), reward.numel(), ws.scalars[6:7])
            L.log('train/batch_reward', ws.scalars[6], step)

        do_cpc = (not self.pixel_sac) and step % self.cpc_update_freq == 0
        if not only_cpc:
            soft = step % self.critic_target_update_freq == 0
            self._soft_update_hint, self._soft_update_done = soft, False
            try:
                self.update_critic(obs, action, reward, next_obs, not_done, L, step, noise=noise_c)
            finally:
                self._soft_update_hint = False
            # The target soft update reads the critic's parameters, which the actor phase does not touch (it steps the
            # actor's own tensors and log_alpha): applied BEFORE the actor phase it gives the same numbers as after
            # it (curl_sac.py:437-445), and the target encoder is then final when the actor phase encodes obs -- so
            # the positives' target pass can share those launches.  (Normally it has already happened: inside the
            # critic's Adam launch.)
            if soft and not self._soft_update_done:
                self.soft_update_targets()
            if step % self.actor_update_freq == 0:
                if do_cpc and isinstance(cpc_kwargs.get("obs_pos"), ObsRef):
                    self._pos_hint = cpc_kwargs["obs_pos"]
                self._defer_actor_step = do_cpc
                try:
                    self.update_actor_and_alpha(obs, L, step, noise=noise_a)
                finally:
                    self._pos_hint = None
                    self._defer_actor_step = False

        if do_cpc:
            obs_anchor, obs_pos = cpc_kwargs["obs_anchor"], cpc_kwargs["obs_pos"]
            self.update_cpc(obs_anchor, obs_pos, cpc_kwargs, L, step)
        self._finish_actor_step()

    # ---------------------------------------------------------------- captured update graphs
    def enable_update_graphs(self, replay_buffer, warm=1, depth=2):
        """Replay ``update()`` from captured hipGraphs (kernels only enqueue, take no host syncs and
        allocate nothing, so a whole update can be captured).  One graph per KIND of step -- (actor phase?, target soft
        update?, CURL phase?): with train.py's frequencies that is "even" and "odd" -- is captured the first time the
        kind comes up after ``warm`` eager updates of it, and replayed from then on: the host's work per update shrinks
        to drawing the indices (NumPy, the reference's stream and order), writing them and ~80 bytes of per-update
        control values into a pinned block, and one graph launch.  What changes from update to update is DATA the
        captured kernels read on the device: the minibatch's indices / crop offsets (as before), the Philox stream
        position of each policy-noise draw and every optimizer's two step-dependent Adam factors (curla_hip.h: the
        ``rng_dev`` / ``dyn`` arguments); the block's staging kernel is the first node of the graph.
        ``depth`` graphs are captured per kind, each with its own pinned block, and used in rotation: the host may then
        prepare update n + 2 depth - 1 while the GPU still reads the block of update n (with one graph per kind it would
        wait for the replay two updates back before every update).
        Results are bit-identical to the eager path (tests/test_gpu_graph.py).  Steps the graphs do not cover run
        eagerly, in any mix: logging steps (``step % log_interval == 0``: they compute extra scalars), histogram /
        image recording steps, ``only_cpc``, float augmentations (ColorJiggle / NoisyCover stage their parameters per
        call), other replay buffers, data-parallel runs on a backend other than RCCL.  Data-parallel updates over RCCL
        ARE captured, collectives included (round 5; the replica check stays on the host, in front of the replay).
        Values the graphs hold as kernel arguments (discount, taus, betas / eps, detach_encoder, the update
        frequencies, the data-parallel group and schedule) are fingerprinted at capture: editing one of them drops the
        graphs and they are captured again; so does ``load_checkpoint``.  A capture that raises leaves the agent's
        state untouched and that update runs eagerly."""
        if self.device.type != "cuda":
            raise RuntimeError("update graphs need the HIP device")
        if not getattr(replay_buffer, "graph_supported", lambda: False)():
            raise ValueError("enable_update_graphs: this replay buffer / augmentation is not graph-replayable "
                             "(uint8-ring minibatches only: RandomCrop or identity, plain storage)")
        opts = (self.critic_optimizer, self.actor_optimizer, self.encoder_optimizer, self.cpc_optimizer)
        if not all(isinstance(o, FlatAdam) and "step" not in vars(o) for o in opts) or \
                type(self.log_alpha_optimizer) is not torch.optim.Adam or self._noise_launch:
            raise ValueError("enable_update_graphs needs the FlatAdam optimizers and the in-kernel policy noise")
        self._graphs = {}
        self._graph_rb = replay_buffer
        self._graph_warm = int(warm)
        self._graph_depth = max(1, int(depth))
        self._graph_seen = {}
        self._graph_key_at_capture = None

    def disable_update_graphs(self):
        self._graphs = None

    def _graph_kind(self, step):
        return (step % self.actor_update_freq == 0, step % self.critic_target_update_freq == 0,
                (not self.pixel_sac) and step % self.cpc_update_freq == 0)

    def _graph_usable(self, replay_buffer, step, only_cpc):
        # data parallel: RCCL collectives are capturable (they are enqueued on streams like kernels); a backend that is
        # staged through the host (gloo) is not
        # (_dp_avg: the group's backend is RCCL; every other backend takes the sum-and-divide / staged path)
        return (replay_buffer is self._graph_rb and not only_cpc and not (self._dp_active and not self._dp_avg)
                and not self._records(step) and step % self.log_interval != 0 and self.training)

    def _graph_live(self, opt):
        """The parameters ``opt``'s step touches in a captured update (None: all).  Under detach_encoder the critic's
        convs have no gradient at step time (curl_sac.py:358): Adam skips them and their step counts stay behind."""
        if opt is self.critic_optimizer and self.detach_encoder:
            convs = {id(p) for m in self.critic.encoder.convs for p in (m.weight, m.bias)}
            return [p for p in opt._plist if id(p) not in convs]
        return None

    def _graph_key(self):
        """Everything a captured graph holds BY VALUE (kernel arguments baked in at capture): a change of any of it
        makes the captured graphs stale -- they are dropped and re-captured (``lr`` travels as data and is not here)."""
        opts = (self.critic_optimizer, self.actor_optimizer, self.encoder_optimizer, self.cpc_optimizer,
                self.log_alpha_optimizer)
        hyper = tuple((float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g.get("weight_decay", 0)))
                      for o in opts for g in o.param_groups)
        return (float(self.discount), float(self.critic_tau), float(self.encoder_tau), bool(self.detach_encoder),
                bool(self.pixel_sac), float(self.target_entropy), float(self.actor.log_std_min),
                float(self.actor.log_std_max), self.actor_update_freq, self.critic_target_update_freq,
                self.cpc_update_freq, self._dp_active, self._dp_overlap, id(self._dp_group) if self._dp_active else 0,
                hyper)

    def _graph_tail(self, kind, B):
        """The 80 control bytes of one graphed update (ReplayBuffer.GRAPH_TAIL) -- and the host-side bookkeeping of
        everything they stand for: the torch generator's offset moves on as _noise() would move it, every optimizer that
        steps in this kind of update counts its step.  Only called once the update's graph exists (a capture that failed
        has run eagerly instead), so there is nothing to take back."""
        do_actor, _, do_cpc = kind
        u64 = np.zeros(4, dtype=np.uint64)
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None
                                            else torch.cuda.current_device()]
        n = B * self.action_dim
        for j in range(2 if do_actor else 1):  # critic-phase draw, then the actor phase's (curl_sac.py:352, 378)
            off = gen.get_offset()
            gen.set_offset(off + 4 * ((n + 3) // 4))
            u64[2 * j], u64[2 * j + 1] = np.uint64(gen.initial_seed() & (2 ** 64 - 1)), np.uint64(off // 4)
        f64 = np.zeros(2, dtype=np.float64)
        f32 = np.zeros(8, dtype=np.float32)
        steps = [(0, self.critic_optimizer)]
        if do_actor:
            steps.append((1, self.actor_optimizer))
            lo = self.log_alpha_optimizer
            g, p = lo.param_groups[0], lo.param_groups[0]["params"][0]
            st = lo.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
            t64 = int(round(float(st["step"]))) + 1
            b1, b2 = float(g["betas"][0]), float(g["betas"][1])
            f64[0], f64[1] = float(g["lr"]) / (1.0 - b1 ** float(t64)), (1.0 - b2 ** float(t64)) ** 0.5
            st["step"] = torch.tensor(float(t64), dtype=torch.float32)
        if do_cpc:
            steps += [(2, self.encoder_optimizer), (3, self.cpc_optimizer)]
        for slot, opt in steps:
            live = self._graph_live(opt)
            f32[2 * slot], f32[2 * slot + 1] = opt.hyper_floats(opt.next_step(live))
            opt.advance(live)
        return u64.tobytes() + f64.tobytes() + f32.tobytes()

    GRAPH_CAPTURE_RETRIES = 3  # failed captures (on any rank) after which update graphs are switched off

    def _graph_capture_agreed(self, ok):
        """True when the capture succeeded on EVERY rank of the data-parallel group (one rank: ``ok`` itself)."""
        if not (self._dp_active and self._dp_world > 1):
            return bool(ok)
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._dp_group)
        return bool(int(flag.item()))

    def _drop_graphs(self):
        """Forget the captured graphs (they are re-captured after ``warm`` further eager updates of each kind)."""
        if self._graphs is not None:
            torch.cuda.synchronize()
            self._graphs = {}
            self._graph_seen = {}
            self._graph_key_at_capture = None

    def _update_graphed(self, rb, L, step):
        key = self._graph_key()
        if self._graphs and key != self._graph_key_at_capture:
            self._drop_graphs()  # a value the graphs hold as a kernel argument has been edited since the capture
        kind = self._graph_kind(step)
        seen = self._graph_seen.get(kind, 0)
        self._graph_seen[kind] = seen + 1
        if seen < self._graph_warm:  # (first uses allocate workspaces and set kernel attributes: not capturable)
            return self._update_eager(rb, L, step)
        ring = self._graphs.setdefault(kind, [])
        turn = (seen - self._graph_warm) % self._graph_depth
        if turn >= len(ring):
            ring.append(dict(slot=sum(len(r) for r in self._graphs.values()), graph=None))
        st = ring[turn]
        B = rb.batch_size
        if st["graph"] is None:
            # capture FIRST, commit the host's bookkeeping (NumPy draw, generator offset, step counts) only once the
            # capture has succeeded: nothing inside the captured region draws from NumPy or moves the generator
            # (draw_indices / _graph_tail run after it), so a capture that raises (a first-use attribute call after an
            # option change, a HIP call from another thread) has touched only the phase-to-phase hints below, which are
            # reset, and this update runs eagerly
            blk = rb.graph_block(st["slot"])
            base = blk["dev"].data_ptr() + blk["tail"]
            dyn = {self.critic_optimizer: base + 48, self.actor_optimizer: base + 56, self.encoder_optimizer: base + 64,
                   self.cpc_optimizer: base + 72}
            self._graph_cap = dict(rng=(base, base + 16), n_noise=0)
            for opt, addr in dyn.items():
                opt._dyn = addr
            self.log_alpha_optimizer._curla_dyn64 = base + 32
            null = _NullLog()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            err = None
            try:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._update_phases(rb.graph_refs(st["slot"]), null, step)
            except Exception as e:  # noqa: BLE001
                err = e
            finally:
                self._graph_cap = None
                for opt in dyn:
                    opt._dyn = None
                self.log_alpha_optimizer._curla_dyn64 = None
            if err is not None:
                # what _update_phases sets from one phase to the next and had no chance to clear
                self._pos_hint = self._anchor_cache = self._pos_cache = None
                self._pos_fc_done = self._soft_update_hint = self._soft_update_done = False
                self._defer_actor_step = self._actor_step_pending = False
                self._dp_pending = []
            # data parallel: every rank captures at the same step; ONE rank replaying while another runs eagerly would
            # still pair up collective for collective, but the ranks would sit on different schedules indefinitely with
            # nothing checking it -- so the outcome is agreed on (one host-side flag, at capture time only)
            ok_everywhere = self._graph_capture_agreed(err is None)
            if not ok_everywhere:
                self._graph_failures = getattr(self, "_graph_failures", 0) + 1
                where = "on this rank" if err is not None else "on another rank"
                give_up = self._graph_failures >= self.GRAPH_CAPTURE_RETRIES
                warnings.warn(f"curla_amd: capturing the update graph of {kind} failed {where} ({err!r}); this update "
                              "runs eagerly" + (" and update graphs are now disabled (every rank of the group does the same)"
                                                if give_up else " and the capture is tried again next time"),
                              RuntimeWarning, stacklevel=3)
                if give_up:
                    self._graphs = None
                else:
                    self._graph_seen[kind] = seen  # (the slot keeps its empty entry: the same turn captures again)
                return self._update_eager(rb, L, step)
            st["graph"] = graph
            self._graph_key_at_capture = key
        idxs, offs = rb.draw_indices()
        tail = self._graph_tail(kind, B)
        blk = rb.graph_write(st["slot"], idxs, offs, tail)
        if self._dp_active and self._dp_check_every > 0 and step % self._dp_check_every == 0:
            self.check_replicas()
        st["graph"].replay()
        if blk["event"] is None:
            blk["event"] = torch.cuda.Event()
        blk["event"].record()

    def save(self, model_dir, augmentation, step):
        """curl_sac.py:453-456 (same three files, reference tensor layouts)."""
        torch.save(self.CURL.state_dict(), '%s/%s_curl_%s.pt' % (model_dir, augmentation, step))
        torch.save(self.actor.state_dict(), '%s/%s_actor_%s.pt' % (model_dir, augmentation, step))
        torch.save(self.critic.state_dict(), '%s/%s_critic_%s.pt' % (model_dir, augmentation, step))

    def load(self, model_dir, augmentation, step):
        """curl_sac.py:458-465 (same files, same three progress lines on stdout)."""
        self.CURL.load_state_dict(torch.load('%s/%s_curl_%s.pt' % (model_dir, augmentation, step)))
        print('Loaded model %s/%s_curl_%s.pt' % (model_dir, augmentation, step))
        self.actor.load_state_dict(torch.load('%s/%s_actor_%s.pt' % (model_dir, augmentation, step)))
        print('Loaded model %s/%s_actor_%s.pt' % (model_dir, augmentation, step))
        self.critic.load_state_dict(torch.load('%s/%s_critic_%s.pt' % (model_dir, augmentation, step)))
        self.critic_target.load_state_dict(self.critic.state_dict())
        print('Loaded model %s/%s_critic_%s.pt' % (model_dir, augmentation, step))

    # -- full training state (the reference only writes the three state_dicts above and
    #    cannot resume; this adds log_alpha, the target critic, the five Adam states, the RNG streams and the step)
    def _optimizers(self):
        return {"actor": self.actor_optimizer, "critic": self.critic_optimizer, "log_alpha": self.log_alpha_optimizer,
                "encoder": self.encoder_optimizer, "cpc": self.cpc_optimizer}

    def _convert_opt_state(self, opt, sd, to_reference):
        """Adam moments of an encoder fc.weight follow the weight's column order: (c,y,x) on disk like the
        reference's Parameter, (y,x,c) in HBM."""
        fc_of = {id(e.fc.weight): e.fc for e in (self.actor.encoder, self.critic.encoder, self.critic_target.encoder)}
        params = [q for g in opt.param_groups for q in g["params"]]
        state = {}
        for i, st in sd["state"].items():
            st = dict(st)
            fc = fc_of.get(id(params[i]))
            for k, v in st.items():
                if torch.is_tensor(v):
                    if fc is not None and v.dim() == 2:
                        v = (fc.to_reference_layout(v) if to_reference else fc.from_reference_layout(v)).contiguous()
                    st[k] = v.detach().cpu() if to_reference else v
            state[i] = st
        return {"state": state, "param_groups": sd["param_groups"]}

    def save_checkpoint(self, path, step):
        rng = {"torch": torch.get_rng_state(), "numpy": np.random.get_state()}
        if self.device.type == "cuda":
            rng["cuda"] = torch.cuda.get_rng_state(self.device)
        torch.save({
            "format": "curla_amd.checkpoint.v1", "step": int(step),
            "actor": self.actor.state_dict(), "critic": self.critic.state_dict(),
            "critic_target": self.critic_target.state_dict(), "curl": self.CURL.state_dict(),
            "log_alpha": self.log_alpha.detach().cpu(),
            "optimizers": {k: self._convert_opt_state(o, o.state_dict(), True) for k, o in self._optimizers().items()},
            "rng": rng,
        }, path)

    def load_checkpoint(self, path, restore_rng=True):
        """Returns the step the checkpoint was written at."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if ck.get("format") != "curla_amd.checkpoint.v1":
            raise ValueError("not a curla_amd checkpoint: %r" % (path,))
        self.CURL.load_state_dict(ck["curl"])
        self.actor.load_state_dict(ck["actor"])
        self.critic.load_state_dict(ck["critic"])
        self.critic_target.load_state_dict(ck["critic_target"])
        with torch.no_grad():
            self.log_alpha.copy_(ck["log_alpha"])
        for k, o in self._optimizers().items():
            o.load_state_dict(self._convert_opt_state(o, ck["optimizers"][k], False))
        if restore_rng:
            torch.set_rng_state(ck["rng"]["torch"])
            np.random.set_state(ck["rng"]["numpy"])
            if self.device.type == "cuda" and "cuda" in ck["rng"]:
                torch.cuda.set_rng_state(ck["rng"]["cuda"], self.device)
        self._anchor_cache = None
        # captured graphs hold the addresses of log_alpha's Adam moments, which load_state_dict has just replaced
        self._drop_graphs()
        return ck["step"]

