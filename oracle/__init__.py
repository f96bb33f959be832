'''CPU oracle for the flexdiffuse image-guided denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's
algorithm (tim-speed/flexdiffuse) used as the *checker* for the HIP path.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it; the product package `flexdiffuse_amd` never does and fails loudly when its HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * `oracle.guidance_ref`  -- PINNED: checked against golden vectors produced by
    importing the reference's own `guidance.py` in the build container
    (`tests/golden/make_guidance_goldens.py` -> `tests/golden/guidance_*.npz`).
  * `oracle.clip_ref` (preprocess / CLIP towers) -- PINNED on a tiny seeded CLIP
    config against the reference's `encode/clip.py` driving `transformers`
    (`tests/golden/make_clip_goldens.py`).
  * `oracle.unet_ref`, `oracle.vae_ref`, `oracle.ddim_ref`, `oracle.pipeline_ref`
    -- PARITY UNPINNED: the arithmetic lives in diffusers==0.3.0, which is not
    vendored in the reference and not installed anywhere here.  These restate the
    published architecture, anchored by exact parameter
    counts, analytic DDIM tables and the reference's own call sites
    (`pipeline/flex.py`, `pipeline/guide.py`).
'''
