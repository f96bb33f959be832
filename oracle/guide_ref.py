'''CPU oracle: `Guide.embeds` control flow (guidance.py:337-474) on top of the CLIP and
guidance oracles.  TEST INFRASTRUCTURE ONLY.  PINNED by tests/golden/clip_goldens.npz
(`guide/*` arrays captured from the reference's own Guide on a tiny seeded CLIP).
Batched prompts with a guide are tweened row by row (the reference raises there, E2).
'''
import torch

from . import clip_ref, guidance_ref as G


class GuideRef():
    def __init__(self, sd, cfg, tokenizer):
        self.sd, self.cfg, self.tok = sd, cfg, tokenizer
        self.placeholder = self.prompt('{}')

    def prompt(self, p):
        ids = self.tok(p, padding='max_length', max_length=self.tok.model_max_length,
                       truncation=True, return_tensors='pt').input_ids
        return clip_ref.text_hidden(self.sd, self.cfg, ids)

    def image(self, img):
        return clip_ref.image_tokens(self.sd, self.cfg,
                                     clip_ref.clip_pixels(clip_ref.preprocess(img)))

    def embeds(self, prompt='', guide=None, mapping_concepts='', guide_threshold_mult=0.5,
               guide_threshold_floor=0.5, guide_clustered=0.5, guide_linear=(0.0, 0.5),
               guide_max_guidance=0.5, guide_header_max=0.15, guide_mode=1, guide_reuse=True):
        if isinstance(prompt, str):
            prompt = prompt.strip()
        elif isinstance(prompt, list):
            prompt = [s.strip() for s in prompt if s.strip()]
        else:
            raise ValueError('`prompt` has to be of type `str` or `list`')
        if not prompt and guide is None:
            raise ValueError('No prompt, or guide image provided.')
        text = self.prompt(prompt) if prompt else None
        g_emb, concept = None, None
        if guide is not None:
            if isinstance(guide, str):
                guide = guide.strip()
                if guide:
                    g_emb = self.prompt(guide)
            else:
                g_emb = self.image(guide)
                if mapping_concepts:
                    concept = self.prompt(mapping_concepts)
        if text is not None:
            if g_emb is None:
                return text
            rows = []
            for b in range(text.shape[0]):
                out, _, _ = G.tween(text[b], g_emb, (guide_threshold_floor, guide_threshold_mult),
                                    guide_linear, guide_clustered, guide_max_guidance,
                                    guide_header_max, guide_mode, guide_reuse)
                if concept is not None:
                    out = G.concept_override(g_emb, concept, text[b], out=out)
                rows.append(out)
            return torch.cat(rows)
        if isinstance(guide, str):
            return g_emb
        out = g_emb[:, :self.tok.model_max_length, :].clone()
        out[:, 0, :] += (self.placeholder[:, 0, :] - out[:, 0, :]) * 0.85
        return out
