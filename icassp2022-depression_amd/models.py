"""The three model families of the hot path, each a forward + explicit backward over libdep_rnn.so.

  AudioGRU   -- `AudioBiLSTM` of Classification/audio_gru_whole.py:24-108 (variant 'clf': LayerNorm,
                mean pool, Softmax) and Regression/audio_bilstm_perm.py:45-127 ('reg': no LN, sum pool, ReLU)
  TextBiLSTM -- Classification/text_bilstm_whole.py:23-114 ('clf') / Regression/text_bilstm_perm.py:37-124 ('reg')
  FusionNet  -- `fusion_net` + `MyLoss` of Classification/fuse_net_whole.py:245-395 ('clf') and
                Regression/fuse_net.py:224-366 ('reg')

Parameter names, shapes and order reproduce the reference's state_dict() (they are API: the fusion
script transplants weights by name, fuse_net_whole.py:566-588).  Dropout masks come from Philox keyed
by a per-forward seed; `eval()` disables them like nn.Dropout / nn.GRU(dropout=...).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import nn
from . import parallel


def _rnn_names(prefix, F, H, Lyr, dirs, G):
    out = []
    for l in range(Lyr):
        for d in range(dirs):
            s = f'l{l}' + ('_reverse' if d else '')
            inp = F if l == 0 else H * dirs
            out += [(f'{prefix}.weight_ih_{s}', (G * H, inp)), (f'{prefix}.weight_hh_{s}', (G * H, H)),
                    (f'{prefix}.bias_ih_{s}', (G * H,)), (f'{prefix}.bias_hh_{s}', (G * H,))]
    return out


class _RnnCache:
    """dep_rnn descriptors + reserve/workspace per (B,T,training) so steady-state steps never allocate."""

    def __init__(self, cell, F, H, Lyr, dirs, p, pool, device):
        self.args = (cell, F, H, Lyr, dirs, p, pool, device)
        self.cache = {}

    def get(self, B, T, training):
        key = (B, T, bool(training))
        r = self.cache.get(key)
        if r is None:
            cell, F, H, Lyr, dirs, p, pool, device = self.args
            if len(self.cache) > 6:
                self.cache.clear()
            r = L.Rnn(cell, B, T, F, H, Lyr, dirs, training, p if training else 0.0, pool, device)
            self.cache[key] = r
        self.last = r
        return r

    def check(self):
        """dep_rnn_status of the most recently used stack (synchronises; raises DepError if a sweep gave up)."""
        if getattr(self, 'last', None) is not None:
            self.last.check()

    def status_word(self):
        return self.last.status_word() if getattr(self, 'last', None) is not None else None

    def fallback_word(self):
        return self.last.fallback_word() if getattr(self, 'last', None) is not None else None


def _mlp_feature(x, W1, b1, p, seed, sites):
    """Dropout(p) -> Linear -> ReLU -> Dropout(p), forward only (the frozen encoders' heads in fusion_net.pretrained_feature):
    one launch when csrc/head.hip covers the widths, the three composed launches otherwise."""
    if x.is_contiguous() and W1.is_contiguous() and L.head_mlp_supported(W1.shape[1], W1.shape[0], 0):
        first = p > 0
        z1 = torch.empty(x.shape[0], W1.shape[0], dtype=torch.float32, device=x.device); a1 = torch.empty_like(z1)
        L.head_mlp_fwd(x, W1, b1, None, None, torch.empty_like(x) if first else None, z1, a1, None, p, seed, sites, first)
        return a1
    a0 = x
    if p > 0:
        a0 = torch.empty_like(x); L.dropout(x, a0, p, seed, sites[0])
    z = L.linear_fwd(a0, W1, b1)
    a1 = torch.empty_like(z); L.relu_dropout_fwd(z, a1, p, seed, sites[1])
    return a1


class _MLPHead:
    """[Dropout] -> Linear(H,H) -> ReLU -> Dropout -> [Linear(H,C)]   (fc_audio / fc_out Sequentials)."""

    def __init__(self, owner, n1, n2, p, first_dropout, sites):
        self.o = owner; self.n1 = n1; self.n2 = n2; self.p = p; self.first = first_dropout; self.sites = sites

    def forward(self, x, seed, training):
        o = self.o; P = o._params
        p = self.p if training else 0.0
        W1, b1 = P[self.n1 + '.weight'].data, P[self.n1 + '.bias'].data
        W2 = P[self.n2 + '.weight'].data if self.n2 is not None else None
        if x.is_contiguous() and W1.is_contiguous() and L.head_mlp_supported(W1.shape[1], W1.shape[0], 0 if W2 is None else W2.shape[0]):
            # one launch (csrc/head.hip): both dropouts, both Linears, the ReLU
            B, H1 = x.shape[0], W1.shape[0]
            first = bool(self.first and p > 0)
            a0 = torch.empty_like(x) if first else x
            z1 = torch.empty(B, H1, dtype=torch.float32, device=x.device); a1 = torch.empty_like(z1)
            z2 = torch.empty(B, W2.shape[0], dtype=torch.float32, device=x.device) if W2 is not None else None
            L.head_mlp_fwd(x, W1, b1, W2, P[self.n2 + '.bias'].data if W2 is not None else None, a0 if first else None,
                           z1, a1, z2, p, seed, self.sites, first)
            self.saved = (a0, z1, a1, p, seed)
            return z2 if W2 is not None else a1
        if self.first and p > 0:
            a0 = torch.empty_like(x); L.dropout(x, a0, p, seed, self.sites[0])
        else:
            a0 = x
        z1 = L.linear_fwd(a0, W1, b1)
        a1 = torch.empty_like(z1)
        L.relu_dropout_fwd(z1, a1, p, seed, self.sites[1])
        z2 = None
        if self.n2 is not None:
            z2 = L.linear_fwd(a1, W2, P[self.n2 + '.bias'].data)
        self.saved = (a0, z1, a1, p, seed)
        return z2 if self.n2 is not None else a1

    def backward(self, dz2):
        """dz2: grad of the second Linear's output (B,C). Returns grad of the head input (B,H)."""
        o = self.o; P = o._params
        a0, z1, a1, p, seed = self.saved
        W1, W2 = P[self.n1 + '.weight'], P[self.n2 + '.weight']
        B, Cc = dz2.shape; H = a1.shape[1]
        if (dz2.is_contiguous() and a0.is_contiguous() and W1.data.is_contiguous() and W1._grad.is_contiguous()
                and L.head_mlp_supported(a0.shape[1], H, Cc)):
            dx = torch.empty_like(a0); dz1 = torch.empty_like(z1)
            L.head_mlp_bwd(dz2, a0, z1, a1, W1.data, W2.data, W1._grad, P[self.n1 + '.bias']._grad, W2._grad,
                           P[self.n2 + '.bias']._grad, dx, dz1, p, seed, self.sites, bool(self.first and p > 0))
            return dx
        L.gemm(1, 0, Cc, H, B, dz2, Cc, a1, H, W2._grad, H)                       # dW2 = dz2^T a1
        L.colsum(dz2, P[self.n2 + '.bias']._grad)
        da1 = torch.empty_like(a1)
        L.gemm(0, 0, B, H, Cc, dz2, Cc, W2.data, H, da1, H)                        # da1 = dz2 W2
        dz1 = torch.empty_like(z1)
        L.relu_dropout_bwd(da1, z1, dz1, p, seed, self.sites[1])
        Hin = a0.shape[1]
        L.gemm(1, 0, H, Hin, B, dz1, H, a0, Hin, W1._grad, Hin)                    # dW1 = dz1^T a0
        L.colsum(dz1, P[self.n1 + '.bias']._grad)
        dx = torch.empty_like(a0)
        L.gemm(0, 0, B, Hin, H, dz1, H, W1.data, Hin, dx, Hin)
        if self.first and p > 0:
            L.dropout(dx, dx, p, seed, self.sites[0])
        return dx


class AudioGRU(nn.Module):
    def __init__(self, config, variant='clf', seed=None):
        super().__init__()
        self.variant = variant
        self.num_classes = config['num_classes']; self.learning_rate = config['learning_rate']
        self.dropout = float(config['dropout']); self.hidden_dims = H = config['hidden_dims']
        self.rnn_layers = Lyr = config['rnn_layers']; self.embedding_size = F = config['embedding_size']
        self.bidirectional = config.get('bidirectional', False)
        if self.bidirectional:
            raise L.DepError('AudioBiLSTM: the reference runs a unidirectional GRU (bidirectional=False)')
        gen = nn.make_generator(seed)
        init = {}
        # definition order of the reference (named_parameters order)
        self._add('attention_layer.0.weight', (H, H), live=False); self._add('attention_layer.0.bias', (H,), live=False)
        init.update(nn.default_linear_init('attention_layer.0.weight', 'attention_layer.0.bias', H, H, gen))
        rn = _rnn_names('lstm_net_audio', F, H, Lyr, 1, 3)
        for n, s in rn:
            self._add(n, s)
        init.update(nn.default_rnn_init(rn, H, gen))
        self._buffers = {}
        if variant == 'clf':
            self._add('ln.weight', (F,)); self._add('ln.bias', (F,))
            init['ln.weight'] = np.ones(F, np.float32); init['ln.bias'] = np.zeros(F, np.float32)
        else:
            self._add('bn.weight', (3,), live=False); self._add('bn.bias', (3,), live=False)
            init['bn.weight'] = np.ones(3, np.float32); init['bn.bias'] = np.zeros(3, np.float32)
            self._buffers = {'bn.running_mean': torch.zeros(3), 'bn.running_var': torch.ones(3),
                             'bn.num_batches_tracked': torch.zeros((), dtype=torch.long)}
        self._add('fc_audio.1.weight', (H, H)); self._add('fc_audio.1.bias', (H,))
        self._add('fc_audio.4.weight', (self.num_classes, H)); self._add('fc_audio.4.bias', (self.num_classes,))
        init.update(nn.default_linear_init('fc_audio.1.weight', 'fc_audio.1.bias', H, H, gen))
        init.update(nn.default_linear_init('fc_audio.4.weight', 'fc_audio.4.bias', self.num_classes, H, gen))
        self._finalize(init)
        self._rnn_w = [self._params[n].data for n, _ in rn]
        self._rnn_g = [self._params[n]._grad for n, _ in rn]
        if variant == 'clf':
            # LayerNorm's affine is folded into gru.weight_ih_l0 / bias_ih_l0 (dep_ln_fold_*): the stack runs on x-hat
            # with the folded pair and hands back their gradients, from which dW, db, dgamma, dbeta follow exactly --
            # no dL/d(xn) contraction and no pass over (B*T, F) for the two LayerNorm parameter gradients.
            mk = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.device)
            self._fold = (mk(3 * H, F), mk(3 * H), mk(3 * H, F), mk(3 * H))          # Wf, bf, dWf, dbf
            self._rnn_w_fold = [self._fold[0], self._rnn_w[1], self._fold[1]] + self._rnn_w[3:]
            self._rnn_g_fold = [self._fold[2], self._rnn_g[1], self._fold[3]] + self._rnn_g[3:]
        pool = L.POOL_MEAN if variant == 'clf' else L.POOL_SUM
        self._rnns = _RnnCache(L.CELL_GRU, F, H, Lyr, 1, self.dropout, pool, self.device)
        self._head = _MLPHead(self, 'fc_audio.1', 'fc_audio.4', self.dropout, True, (L.SITE_FC0, L.SITE_FC1))

    def state_dict(self):
        sd = super().state_dict()
        if self._buffers:
            out = type(sd)()
            for k, v in sd.items():
                out[k] = v
                if k == 'bn.bias':
                    out.update(self._buffers)
            return out
        return sd

    def load_state_dict(self, sd, strict=True):
        sd = {k: v for k, v in sd.items() if k not in self._buffers}
        return super().load_state_dict(sd, strict)

    def check_health(self):
        self._rnns.check()

    def status_words(self):
        w = self._rnns.status_word()
        return [] if w is None else [w]

    def fallback_words(self):
        w = self._rnns.fallback_word()
        return [] if w is None else [w]

    # encoder part shared with FusionNet
    def encode(self, x, training, seed):
        B, T, F = x.shape
        P = self._params
        if self.variant == 'clf':
            xn, _ = L.layernorm_fwd(x.view(B * T, F), None, None, save=False)           # x-hat
            L.ln_fold_fwd(self._rnn_w[0], self._rnn_w[2], P['ln.weight'].data, P['ln.bias'].data, *self._fold[:2])
            weights = self._rnn_w_fold
        else:
            xn, weights = x.view(B * T, F), self._rnn_w
        rnn = self._rnns.get(B, T, training)
        pooled = torch.empty(B, self.hidden_dims, dtype=torch.float32, device=self.device)
        rnn.forward(xn, weights, seed=seed, pooled=pooled)
        return pooled, (x, xn, rnn)

    def forward(self, x):
        x = self._to_dev(x)
        training = self.training
        seed = nn.next_dropout_seed() if training else 0
        pooled, enc = self.encode(x, training, seed)
        z = self._head.forward(pooled, seed, training)
        out = torch.empty_like(z)
        kind = L.LOSS_CE_ON_SOFTMAX if self.variant == 'clf' else L.LOSS_L1_RELU
        L.head_loss(kind, z, None, out, None, None, 1.0)
        self._saved = enc
        return nn.Output(out, self, z)

    def sync_plan(self):
        """Flat layout [ln | l0 | l1 | .. | fc_audio]: the top layer's range ends with the head (final before the stack's
        backward starts); layer 0 of the classifier is final only after dep_ln_fold_bwd, together with the LayerNorm pair that
        sits right in front of it -- ONE contiguous range, so a 2-layer step is two all-reduces."""
        Lyr, px = self.rnn_layers, 'lstm_net_audio'
        in_call, post = {}, []
        for l in range(Lyr):
            first, last = f'{px}.weight_ih_l{l}', ('fc_audio.4.bias' if l == Lyr - 1 else f'{px}.bias_hh_l{l}')
            if l == 0 and self.variant == 'clf':
                post.append(self._span('ln.weight', last))
            else:
                in_call[l] = self._span(first, last)
        return in_call, post

    def backward(self, dz):
        x, xn, rnn = self._saved
        dpool = self._head.backward(dz)
        P = self._params
        in_call, post = self.sync_plan()
        gs = parallel.make_grad_sync(self, in_call)
        if self.variant == 'clf':
            rnn.backward(xn, self._rnn_w_fold, self._rnn_g_fold, dpooled=dpool, dx=None, grad_sync=gs)
            L.ln_fold_bwd(self._rnn_w[0], self._fold[2], self._fold[3], P['ln.weight'].data, P['ln.bias'].data,
                          self._rnn_g[0], self._rnn_g[2], P['ln.weight']._grad, P['ln.bias']._grad)
        else:
            rnn.backward(xn, self._rnn_w, self._rnn_g, dpooled=dpool, dx=None, grad_sync=gs)
        self._grad_ready = True
        parallel.finish_grad_sync(self, in_call, post)


class TextBiLSTM(nn.Module):
    def __init__(self, config, variant='clf', seed=None, head_out=True, fc_idx=None):
        super().__init__()
        self.variant = variant
        self.num_classes = config['num_classes']; self.learning_rate = config['learning_rate']
        self.dropout = float(config['dropout']); self.hidden_dims = H = config['hidden_dims']
        self.rnn_layers = Lyr = config['rnn_layers']; self.embedding_size = F = config['embedding_size']
        self.bidirectional = config.get('bidirectional', True)
        if not self.bidirectional:
            raise L.DepError('TextBiLSTM: the reference runs a bidirectional LSTM (bidirectional=True)')
        i1, i2 = fc_idx if fc_idx else ((0, 3) if variant == 'clf' else (1, 4))
        self._fc = (f'fc_out.{i1}', f'fc_out.{i2}')
        gen = nn.make_generator(seed)
        shapes = [('attention_layer.0.weight', (H, H)), ('attention_layer.0.bias', (H,))]
        rn = _rnn_names('lstm_net', F, H, Lyr, 2, 4)
        shapes += rn
        shapes += [(self._fc[0] + '.weight', (H, H)), (self._fc[0] + '.bias', (H,)),
                   (self._fc[1] + '.weight', (self.num_classes, H)), (self._fc[1] + '.bias', (self.num_classes,))]
        init = {}
        for n, s in shapes:
            self._add(n, s)
            # init_weight(): xavier_uniform on every weight, 0 on every bias (text_bilstm_whole.py:37-43)
            init[n] = nn.xavier_uniform(s, gen) if 'weight' in n else np.zeros(s, np.float32)
        if variant == 'clf':
            for n, s in (('ln1.weight', (F,)), ('ln1.bias', (F,)), ('ln2.weight', (H,)), ('ln2.bias', (H,))):
                self._add(n, s, live=False)
                init[n] = np.ones(s, np.float32) if 'weight' in n else np.zeros(s, np.float32)
        self._finalize(init)
        self._rnn_w = [self._params[n].data for n, _ in rn]
        self._rnn_g = [self._params[n]._grad for n, _ in rn]
        self._rnns = _RnnCache(L.CELL_LSTM, F, H, Lyr, 2, self.dropout, L.POOL_NONE, self.device)
        self._head = _MLPHead(self, self._fc[0], self._fc[1], self.dropout, variant == 'reg', (L.SITE_FC0, L.SITE_FC1))

    def check_health(self):
        self._rnns.check()

    def status_words(self):
        w = self._rnns.status_word()
        return [] if w is None else [w]

    def fallback_words(self):
        w = self._rnns.fallback_word()
        return [] if w is None else [w]

    def sync_plan(self):
        """Flat layout [attention | l0 (both directions) | l1 .. | fc_out]: the attention pair (final before the stack's
        backward) rides with layer 0, the head with the top layer."""
        Lyr, px = self.rnn_layers, 'lstm_net'
        in_call = {}
        for l in range(Lyr):
            first = 'attention_layer.0.weight' if l == 0 else f'{px}.weight_ih_l{l}'
            last = (self._fc[1] + '.bias') if l == Lyr - 1 else f'{px}.bias_hh_l{l}_reverse'
            in_call[l] = self._span(first, last)
        return in_call, []

    def encode(self, x, training, seed):
        B, T, F = x.shape
        P = self._params
        rnn = self._rnns.get(B, T, training)
        h_n = torch.empty(2 * self.rnn_layers, B, self.hidden_dims, dtype=torch.float32, device=self.device)
        rnn.forward(x, self._rnn_w, seed=seed, h_n=h_n)
        out = rnn.layer_output()
        ctx, att = L.attn_fwd(out, h_n, P['attention_layer.0.weight'].data, P['attention_layer.0.bias'].data)
        return ctx, (x, rnn, out, att)

    def forward(self, x):
        """x: (B,T,F) batch-first like the reference's call site (it permutes to time-first internally,
        text_bilstm_whole.py:103; the HIP operator consumes batch-first directly)."""
        x = self._to_dev(x)
        training = self.training
        seed = nn.next_dropout_seed() if training else 0
        ctx, enc = self.encode(x, training, seed)
        z = self._head.forward(ctx, seed, training)
        out = torch.empty_like(z)
        kind = L.LOSS_CE_ON_SOFTMAX if self.variant == 'clf' else L.LOSS_SMOOTHL1_RELU
        L.head_loss(kind, z, None, out, None, None, 1.0)
        self._saved = enc
        self.last_alpha = enc[3][0]
        return nn.Output(out, self, z)

    def backward(self, dz):
        x, rnn, out, att = self._saved
        P = self._params
        dctx = self._head.backward(dz)
        dout, dh_n = L.attn_bwd(dctx, out, P['attention_layer.0.weight'].data, att, 2 * self.rnn_layers,
                                P['attention_layer.0.weight']._grad, P['attention_layer.0.bias']._grad)
        in_call, post = self.sync_plan()
        rnn.backward(x, self._rnn_w, self._rnn_g, dy=dout, dh_n=dh_n, dx=None, grad_sync=parallel.make_grad_sync(self, in_call))
        self._grad_ready = True
        parallel.finish_grad_sync(self, in_call, post)


class FusionNet(nn.Module):
    """Frozen audio-GRU + text-BiLSTM encoders -> concat(text, audio) -> Linear(no bias) head.
    Only `fc_final.0.weight` trains (fuse_net_whole.py:590-593); the encoders run forward only."""

    def __init__(self, text_embed_size, text_hidden_dims, rnn_layers, dropout, num_classes, audio_hidden_dims,
                 audio_embed_size, variant='clf', seed=None):
        super().__init__()
        self.variant = variant
        self.text_embed_size = Ft = text_embed_size; self.audio_embed_size = Fa = audio_embed_size
        self.text_hidden_dims = Ht = text_hidden_dims; self.audio_hidden_dims = Ha = audio_hidden_dims
        self.rnn_layers = Lyr = rnn_layers; self.dropout = float(dropout); self.num_classes = num_classes
        gen = nn.make_generator(seed)
        init = {}

        def lin(nw, nb, o, i, live, bias=True):
            self._add(nw, (o, i), live=live)
            if bias:
                self._add(nb, (o,), live=live)
            init.update(nn.default_linear_init(nw, nb, o, i, gen, bias=bias))

        lin('attention_layer.0.weight', 'attention_layer.0.bias', Ht, Ht, False)
        rt = _rnn_names('lstm_net', Ft, Ht, Lyr, 2, 4)
        for n, s in rt:
            self._add(n, s, live=False)
        init.update(nn.default_rnn_init(rt, Ht, gen))
        lin('fc_out.1.weight', 'fc_out.1.bias', Ht, Ht, False)
        ra = _rnn_names('lstm_net_audio', Fa, Ha, Lyr, 1, 3)
        for n, s in ra:
            self._add(n, s, live=False)
        init.update(nn.default_rnn_init(ra, Ha, gen))
        lin('fc_audio.1.weight', 'fc_audio.1.bias', Ha, Ha, False)
        if variant == 'clf':
            self._add('ln.weight', (Fa,), live=False); self._add('ln.bias', (Fa,), live=False)
            init['ln.weight'] = np.ones(Fa, np.float32); init['ln.bias'] = np.zeros(Fa, np.float32)
        lin('modal_attn.weight', None, Ht + Ha, Ht + Ha, False, bias=False)
        lin('fc_final.0.weight', None, num_classes, Ht + Ha, True, bias=False)
        self._finalize(init)
        self._wt = [self._params[n].data for n, _ in rt]
        self._wa = [self._params[n].data for n, _ in ra]
        self._rnn_t = _RnnCache(L.CELL_LSTM, Ft, Ht, Lyr, 2, self.dropout, L.POOL_NONE, self.device)
        self._rnn_a = _RnnCache(L.CELL_GRU, Fa, Ha, Lyr, 1, self.dropout, L.POOL_SUM, self.device)
        self.fc_final = [self._params['fc_final.0.weight']]       # `model.fc_final[0].weight`-style access
        self._params['fc_final.0.weight'].weight = self._params['fc_final.0.weight']

    def check_health(self):
        self._rnn_t.check(); self._rnn_a.check()

    def status_words(self):
        return [w for w in (self._rnn_t.status_word(), self._rnn_a.status_word()) if w is not None]

    def fallback_words(self):
        return [w for w in (self._rnn_a.fallback_word(),) if w is not None]

    def _split(self, x):
        """Accept the reference's list of (audio_i, text_i) pairs or an (audio, text) pair of arrays."""
        if isinstance(x, (tuple, list)) and len(x) == 2 and hasattr(x[0], 'shape') and len(x[0].shape) == 3:
            return self._to_dev(x[0]), self._to_dev(x[1])
        xa = np.stack([np.asarray(e[0]) for e in x]); xt = np.stack([np.asarray(e[1]) for e in x])
        return self._to_dev(xa), self._to_dev(xt)

    def pretrained_feature(self, x):
        """fusion_net.pretrained_feature (fuse_net_whole.py:336-366): no gradients; Dropout stays active in
        train() mode exactly as in the reference."""
        xa, xt = self._split(x)
        training = self.training
        seed = nn.next_dropout_seed() if training else 0
        p = self.dropout if training else 0.0
        P = self._params
        B, T, _ = xt.shape
        # text encoder (inference-mode kernels; dropout sites drawn when training)
        rnn = self._rnn_t.get(B, T, training)
        h_n = torch.empty(2 * self.rnn_layers, B, self.text_hidden_dims, dtype=torch.float32, device=self.device)
        rnn.forward(xt, self._wt, seed=seed, h_n=h_n)
        ctx, _ = L.attn_fwd(rnn.layer_output(), h_n, P['attention_layer.0.weight'].data, P['attention_layer.0.bias'].data)
        tf = _mlp_feature(ctx, P['fc_out.1.weight'].data, P['fc_out.1.bias'].data, p, seed, (L.SITE_FC0, L.SITE_FC1))
        # audio encoder
        Ba, Ta, Fa = xa.shape
        if self.variant == 'clf':
            xn, _ = L.layernorm_fwd(xa.view(Ba * Ta, Fa), P['ln.weight'].data, P['ln.bias'].data, save=False)
        else:
            xn = xa.view(Ba * Ta, Fa)
        rna = self._rnn_a.get(Ba, Ta, training)
        pooled = torch.empty(Ba, self.audio_hidden_dims, dtype=torch.float32, device=self.device)
        rna.forward(xn, self._wa, seed=seed + 1, pooled=pooled)
        af = _mlp_feature(pooled, P['fc_audio.1.weight'].data, P['fc_audio.1.bias'].data, p, seed, (L.SITE_FC2, L.SITE_FC3))
        return tf, af

    def forward(self, x):
        """x: concat(text_feature, audio_feature) (B, Ht+Ha) -> probabilities / score."""
        x = x.data if isinstance(x, nn.Output) else x
        x = x.contiguous()
        P = self._params
        W = P['fc_final.0.weight'].data
        B = x.shape[0]
        if self.variant == 'clf':
            z = L.linear_fwd(x, W, None)
            out = torch.empty_like(z)
            L.head_loss(L.LOSS_CE_LOGITS, z, None, out, None, None, 1.0)      # Softmax(dim=1)
        else:
            g = L.linear_fwd(x, P['modal_attn.weight'].data, None)
            gx = torch.empty_like(x); L.sigmoid_gate(g, x, gx)
            z = L.linear_fwd(gx, W, None)
            out = torch.empty_like(z)
            L.head_loss(L.LOSS_L1_RELU, z, None, out, None, None, 1.0)        # ReLU
        return nn.Output(out, None, z)


class MyLoss:
    """Split-weight loss (fuse_net_whole.py:376-395 ; Regression/fuse_net.py:353-366):
    loss(text_feature W[:, :Ht]^T, y) + loss(audio_feature W[:, Ht:]^T, y), gradient to W only."""

    def __init__(self, variant='clf'):
        self.variant = variant

    def __call__(self, text_feature, audio_feature, target, model):
        Wp = model._params['fc_final.0.weight']
        W = Wp.data
        Cc, D = W.shape
        Ht = text_feature.shape[1]; Ha = audio_feature.shape[1]
        B = text_feature.shape[0]
        dev = W.device
        train = model.training and Wp.requires_grad
        zt = torch.empty(B, Cc, dtype=torch.float32, device=dev); za = torch.empty_like(zt)
        L.gemm(0, 1, B, Cc, Ht, text_feature, Ht, W, D, zt, Cc)
        L.gemm(0, 1, B, Cc, Ha, audio_feature, Ha, W[:, Ht:], D, za, Cc)
        if self.variant == 'clf':
            t = torch.as_tensor(np.asarray(target) if not torch.is_tensor(target) else target)
            nn._check_labels(t, Cc)
            kind = L.LOSS_CE_LOGITS; norm_local = B
            if t.dtype == torch.int64:
                t = t.to(device=dev).contiguous().view(-1); kind |= L.LOSS_LABELS_I64
            else:
                t = t.to(device=dev, dtype=torch.int32).contiguous().view(-1)
        else:
            t = torch.as_tensor(np.asarray(target, dtype=np.float32) if not torch.is_tensor(target) else target)
            t = t.to(device=dev, dtype=torch.float32).contiguous().view(B, Cc)
            kind = L.LOSS_SMOOTHL1; norm_local = B * Cc
        norm = parallel.global_count(norm_local) if train else norm_local
        rows = torch.empty(B, dtype=torch.float32, device=dev)
        val = torch.empty(1, dtype=torch.float32, device=dev)                       # overwritten by the first dep_reduce_loss
        dzt = torch.empty_like(zt) if train else None
        dza = torch.empty_like(za) if train else None
        L.head_loss(kind, zt, t, None, rows, dzt, norm); L.reduce_loss(rows, norm, val)
        L.head_loss(kind, za, t, None, rows, dza, norm); L.reduce_loss(rows, norm, val, accumulate=True)

        def bw():
            g = Wp._grad
            L.gemm(1, 0, Cc, Ht, B, dzt, Cc, text_feature, Ht, g, D)               # dW[:, :Ht] = dzt^T text
            L.gemm(1, 0, Cc, Ha, B, dza, Cc, audio_feature, Ha, g[:, Ht:], D)      # dW[:, Ht:] = dza^T audio
            model._grad_ready = True
            parallel.finish_grad_sync(model, *model.sync_plan())
        return nn.Loss(val, bw if train else None, reduce=train and parallel.world_size() > 1,
                       health=getattr(model, 'check_health', None))
