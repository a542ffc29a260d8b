"""BaseManager / TrainerManager — same method surface as the reference's managers/base_manager.py and
managers/trainer_manager.py, driving the HIP-backed SRModel."""
import torch

from . import ops
from .sr_model import SRModel


class BaseManager:
    def __init__(self, opt, create_model=True):
        self.opt = opt
        if create_model:
            self.create_model(opt)

    def create_model(self, opt):
        """base_manager.py:15-23.  The reference wraps SRModel in DataParallelWithCallback (one process, one thread
        per GPU); here parallelism is one process per GPU + RCCL gradient all-reduce (deepsee_amd.parallel), so the
        wrapped and the unwrapped model are the same object."""
        self.sr_model = SRModel(opt)
        self.sr_model_on_one_gpu = self.sr_model

    def use_gpu(self):
        return True

    def preprocess(self, data, from_dataloader=False):
        """base_manager.py:28-66 + data/preprocessor.py: label -> integer map (kept as uint8 instead of a one-hot
        tensor), HR -> LR bicubic + clamp, everything NHWC on the device."""
        if isinstance(data.get("input_semantics"), ops.Labels):
            return data        # already native: a batch of deepsee_amd.data.DeviceLoader
        if from_dataloader and data["image"].dtype == torch.uint8:
            from .data import device_preprocess      # uint8 wire format: [N,H,W] labels, [N,H,W,3] images
            return device_preprocess(self.opt, data)
        out = dict(data)
        for k, v in data.items():
            if isinstance(v, torch.Tensor) and not v.is_cuda:
                out[k] = v.cuda(non_blocking=True)
        if not from_dataloader:
            return out
        opt = self.opt
        image = ops.to_nhwc(out["image"])
        res = {
            "input_semantics": ops.Labels(ops.label_to_u8(out["label"]), opt.label_nc),
            "image_lr": ops.bicubic_down(image, opt.start_size),
            "image_hr": image,
        }
        if opt.guiding_style_image:
            res["guiding_image"] = ops.to_nhwc(out["guiding_image"])
            res["guiding_label"] = ops.Labels(ops.label_to_u8(out["guiding_label"]), opt.label_nc)
        return res


class TrainerManager(BaseManager):
    """trainer_manager.py:6-96."""

    def __init__(self, opt):
        super().__init__(opt, create_model=True)
        assert opt.isTrain
        self.optimizer_G, self.optimizer_D = self.sr_model_on_one_gpu.create_optimizers(opt)
        self.old_lr = opt.lr
        self.generated = None
        self.logs = {}
        self.g_losses, self.d_losses = {}, {}

    def get_logs(self):
        return {**self.logs, **self.sr_model_on_one_gpu.get_logs()}

    def preprocess_input(self, data):
        return super().preprocess(data, from_dataloader=True)

    def run_generator_one_step(self, data):
        self.optimizer_G.zero_grad()
        d = self.preprocess_input(data)
        g_losses, generated = self.sr_model(d, mode="generator")
        g_loss = sum(g_losses.values()).mean()
        g_loss.backward()
        self.optimizer_G.step(clip=self.opt.gradient_clip)
        self.g_losses = g_losses
        self.generated = generated

    def run_discriminator_one_step(self, data):
        self.optimizer_D.zero_grad()
        d = self.preprocess_input(data)
        d_losses = self.sr_model(d, mode="discriminator")
        d_loss = sum(d_losses.values()).mean()
        d_loss.backward()
        self.optimizer_D.step(clip=self.opt.gradient_clip)
        self.d_losses = d_losses

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def get_latest_generated(self):
        return self.generated

    def save(self, epoch):
        self.sr_model_on_one_gpu.save(epoch)

    def update_learning_rate(self, epoch):
        """trainer_manager.py:76-96: linear decay after opt.niter, TTUR split kept."""
        if epoch > self.opt.niter:
            new_lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        else:
            new_lr = self.old_lr
        if new_lr != self.old_lr:
            if self.opt.no_TTUR:
                new_lr_g, new_lr_d = new_lr, new_lr
            else:
                new_lr_g, new_lr_d = new_lr / 2, new_lr * 2
            for g in self.optimizer_D.param_groups:
                g["lr"] = new_lr_d
            for g in self.optimizer_G.param_groups:
                g["lr"] = new_lr_g
            print("update learning rate: %f -> %f" % (self.old_lr, new_lr))
            self.old_lr = new_lr
