"""Pix2PixTrainer (reference trainers/pix2pix_trainer.py:8-88): owns the model and the two optimizers,
runs one G step / one D step, LR decay, save.  Multi-GPU: when torch.distributed is initialised the
flat gradient arenas are sum-all-reduced (RCCL) between backward and the Adam launch."""
from .distributed import FlatGradSync, broadcast_flat
from .pix2pix_model import Pix2PixModel


class Pix2PixTrainer:
    def __init__(self, opt):
        self.opt = opt
        self.pix2pix_model = Pix2PixModel(opt)
        self.pix2pix_model_on_one_gpu = self.pix2pix_model
        self.generated = None
        self.g_losses, self.d_losses = {}, {}
        if opt.isTrain:
            self.optimizer_G, self.optimizer_D = self.pix2pix_model_on_one_gpu.create_optimizers(opt)
            self.old_lr = opt.lr
            self.sync_G = FlatGradSync(self.optimizer_G.flat_g)
            self.sync_D = FlatGradSync(self.optimizer_D.flat_g)
            broadcast_flat(self.optimizer_G.flat_p)          # identical replicas at step 0
            broadcast_flat(self.optimizer_D.flat_p)

    def run_generator_one_step(self, data):
        self.pix2pix_model.train()
        self.optimizer_G.zero_grad()
        g_losses, generated = self.pix2pix_model(data, mode='generator')
        g_loss = sum(g_losses.values()).mean()
        g_loss.backward()
        self.optimizer_G.step(grad_scale=self.sync_G.all_reduce())
        self.g_losses = g_losses
        self.generated = generated

    def run_discriminator_one_step(self, data):
        self.pix2pix_model.train()
        self.optimizer_D.zero_grad()
        d_losses = self.pix2pix_model(data, mode='discriminator')
        d_loss = sum(d_losses.values()).mean()
        d_loss.backward()
        self.optimizer_D.step(grad_scale=self.sync_D.all_reduce())
        self.d_losses = d_losses

    def get_latest_losses(self, include_log_losses=False):
        losses = {**self.g_losses, **self.d_losses}
        if include_log_losses:
            losses = {**losses, **self.pix2pix_model_on_one_gpu.get_loss_log()}
            self.pix2pix_model_on_one_gpu.reset_loss_log()
        return losses

    def get_latest_generated(self):
        return self.generated

    def save(self, epoch):
        self.pix2pix_model_on_one_gpu.save(epoch)

    def update_learning_rate(self, epoch):
        """Constant for niter epochs, then linear to 0 over niter_decay, keeping TTUR's 1/2 : 2 ratio
        (pix2pix_trainer.py:68-88)."""
        if epoch > self.opt.niter:
            new_lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        else:
            new_lr = self.old_lr
        if new_lr != self.old_lr:
            new_lr_G, new_lr_D = (new_lr, new_lr) if self.opt.no_TTUR else (new_lr / 2, new_lr * 2)
            for g in self.optimizer_D.param_groups:
                g['lr'] = new_lr_D
            for g in self.optimizer_G.param_groups:
                g['lr'] = new_lr_G
            print('update learning rate: %f -> %f' % (self.old_lr, new_lr))
            self.old_lr = new_lr
