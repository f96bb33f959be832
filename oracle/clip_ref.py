'''CPU oracle: CLIP text / vision towers and image preprocessing in torch fp32.

TEST INFRASTRUCTURE ONLY.  Follows the reference's encode/clip.py (preprocess :15-39,
CLIPEncoder.prompt :47-65 = text_model(ids)[0], CLIPEncoder.image :67-100 = all 257
tokens through post_layernorm + visual_projection) and the published CLIP architecture
(transformers `CLIPModel`).  PINNED on a tiny seeded config against
the reference's encode/clip.py driving transformers' CLIPModel
(tests/golden/make_clip_goldens.py -> tests/golden/clip_goldens.npz).
'''
import numpy as np
import torch
import torch.nn.functional as F

from .unet_ref import _lin, attention

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def sd_size(w, h, max_dim=512):
    '''encode/clip.py:24-33.'''
    if h > w:
        return (int(w / (h / max_dim)) // 64) * 64, max_dim
    if w > h:
        return max_dim, (int(h / (w / max_dim)) // 64) * 64
    return max_dim, max_dim


def preprocess(image) -> torch.Tensor:
    '''encode/clip.py:15-39: PIL -> (1,3,H,W) fp32 in [-1,1].'''
    from PIL import Image
    w, h = sd_size(*image.size)
    lanczos = getattr(Image, 'LANCZOS', None) or Image.Resampling.LANCZOS
    arr = np.array(image.resize((w, h), resample=lanczos).convert('RGB')).astype(np.float32)
    t = torch.from_numpy((arr / 255.0)[None].transpose(0, 3, 1, 2).copy())
    return 2.0 * t - 1.0


def clip_pixels(x: torch.Tensor, size: int = 224) -> torch.Tensor:
    '''encode/clip.py:76-84: centre crop, antialiased bicubic resize, CLIP normalise
    (applied to the [-1,1] tensor: reference quirk E4).'''
    H, W = x.shape[-2:]
    c = min(H, W)
    top, left = int(round((H - c) / 2.0)), int(round((W - c) / 2.0))
    x = x[..., top:top + c, left:left + c]
    x = F.interpolate(x, size=(size, size), mode='bicubic', align_corners=False, antialias=True)
    return (x - torch.tensor(CLIP_MEAN).view(-1, 1, 1)) / torch.tensor(CLIP_STD).view(-1, 1, 1)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], 1e-5)


def _act(x, kind):
    if kind == 'quick_gelu':
        return x * torch.sigmoid(1.702 * x)
    return F.gelu(x)


def _encoder(sd, prefix, h, n_layers, heads, act, mask=None):
    for i in range(n_layers):
        p = f'{prefix}.encoder.layers.{i}'
        x = _ln(sd, p + '.layer_norm1', h)
        q, k, v = (_lin(sd, f'{p}.self_attn.{n}', x) for n in ('q_proj', 'k_proj', 'v_proj'))
        h = h + _lin(sd, p + '.self_attn.out_proj', attention(q, k, v, heads, mask))
        x = _ln(sd, p + '.layer_norm2', h)
        h = h + _lin(sd, p + '.mlp.fc2', _act(_lin(sd, p + '.mlp.fc1', x), act))
    return h


@torch.no_grad()
def text_hidden(sd, cfg, ids: torch.Tensor) -> torch.Tensor:
    '''input ids (B,77) -> last_hidden_state (B,77,D) (causal mask, final layer norm).'''
    t = cfg.text
    L = ids.shape[1]
    h = sd['text_model.embeddings.token_embedding.weight'][ids] + \
        sd['text_model.embeddings.position_embedding.weight'][:L][None]
    mask = torch.full((L, L), float('-inf')).triu(1)
    h = _encoder(sd, 'text_model', h, t.num_hidden_layers, t.num_attention_heads, t.hidden_act,
                 mask)
    return _ln(sd, 'text_model.final_layer_norm', h)


@torch.no_grad()
def image_tokens(sd, cfg, pixels: torch.Tensor) -> torch.Tensor:
    '''(B,3,224,224) -> (B,257,proj): every token through post-LN and the projection.'''
    v = cfg.vision
    x = F.conv2d(pixels.float(), sd['vision_model.embeddings.patch_embedding.weight'],
                 stride=v.patch_size)
    B = x.shape[0]
    x = x.flatten(2).transpose(1, 2)
    cls = sd['vision_model.embeddings.class_embedding'].expand(B, 1, -1)
    h = torch.cat([cls, x], dim=1) + sd['vision_model.embeddings.position_embedding.weight'][None]
    h = _ln(sd, 'vision_model.pre_layrnorm', h)
    h = _encoder(sd, 'vision_model', h, v.num_hidden_layers, v.num_attention_heads, v.hidden_act)
    h = _ln(sd, 'vision_model.post_layernorm', h)
    return F.linear(h, sd['visual_projection.weight'])
