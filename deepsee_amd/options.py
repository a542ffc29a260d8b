"""Flat option namespace of the path (the reference threads an argparse.Namespace `opt` through every
constructor; field list: options/base_options.py, options/train_options.py) and the
name-keyed presets of options/configurations.py:3-43."""
from types import SimpleNamespace

DEFAULTS = dict(
    name="deepsee_amd", gpu_ids=[0], model="sr",
    norm_G="spectrallateseansyncbatch3x3", norm_D="spectralinstance", norm_E="spectralinstance",
    add_noise=True, noisy_style_scale=0.2, noisy_style_dist="uniform",
    batchSize=8, load_size=256, crop_size=256, start_size=32, aspect_ratio=1.0,
    label_nc=19, contain_dontcare_label=False, semantic_nc=19, output_nc=3,
    max_fm_size=256, downsampling_method="bicubic",
    netG="deepsee", netE="combinedstyle", netD="multiscale", netD_subarch="n_layer",
    ngf=32, nef=32, ndf=32, num_D=2, n_layers_D=4,
    init_type="xavier", init_variance=0.02, regional_style_size=128,
    full_style_image=False, guiding_style_image=False, random_style_matrix=False,
    model_parallel_mode=0, isTrain=True, continue_train=False, which_epoch="latest",
    beta1=0.0, beta2=0.9, no_TTUR=False, efficient=False, lr=2e-4,
    lambda_feat=10.0, lambda_vgg=10.0, no_ganFeat_loss=False, no_vgg_loss=False,
    gan_mode="hinge", gradient_clip=-1.0, num_upsampling_layers="normal",
    niter=50, niter_decay=25, gpu_info=False, checkpoints_dir="./checkpoints",
    seed=0,
    # build-only options (no counterpart in the reference's parser)
    vgg_weights=None,        # path of torchvision's vgg19 state dict (the reference downloads it, architecture.py:154)
    sync_bn=False,           # data-parallel: BatchNorm statistics over the global batch; default sync-free
    sync_bn_clamp=True,      # ... with the reference DP branch's clamp(var, eps) (batchnorm.py:145) instead of var + eps
    preprocess_mode="resize_and_crop", no_flip=False,
    hip_graphs=True,         # capture the G and the D step as hipGraphs (one per encoder-branch variant and batch shape) and
                             # replay them: 0.5 ms instead of ~60 ms of Python launch enqueue per step
    max_graph_shapes=4,      # ... for at most this many distinct batch shapes (least recently used evicted)
    dp_comm="torch",         # data parallel collectives: "torch" (torch.distributed, backend nccl = RCCL) | "capi" (dsee_comm_*)
    dp_graph_collectives=False,  # data parallel + hip_graphs: ALSO capture the chunked RCCL all-reduce and per-chunk Adam (off:
                                 # capturing RCCL operations aborts intermittently in the HIP runtime on ROCm 7.0.2 / RCCL 2.26.6)
    precision="fp32",        # "fp16": one-term scaled-fp16 matrix-core GEMMs + fp16 Winograd-domain products (BASELINE configs[2])
    kernel_plan=None,        # deepsee_amd.plan.KernelPlan, or a dict of its field overrides: which equivalent kernel paths the model takes
)

PRESETS = {
    # options/configurations.py: independent vs guided, 8x 32->256 and 32x 16->512
    "independent_8x_256": dict(netE="combinedstyle", noisy_style_scale=0.2, start_size=32, crop_size=256,
                               load_size=256, add_noise=True, max_fm_size=256),
    "guided_8x_256": dict(netE="fullstyle", noisy_style_scale=0.05, guiding_style_image=True, start_size=32,
                          crop_size=256, load_size=256, add_noise=True, max_fm_size=256),
    "independent_32x_512": dict(netE="combinedstyle", noisy_style_scale=0.2, start_size=16, crop_size=512,
                                load_size=512, add_noise=False, max_fm_size=256),
    "independent_8x_32": dict(netE="combinedstyle", noisy_style_scale=0.2, start_size=4, crop_size=32, load_size=32,
                              add_noise=True, max_fm_size=256),
}


def make_opt(preset=None, **over):
    d = dict(DEFAULTS)
    if preset:
        d.update(PRESETS[preset])
    d.update(over)
    return SimpleNamespace(**d)
