"""Minimal Gaussian parameter container exposing exactly the getter contract the tracer consumes
(reference threedgrut/model/model.py:45-205, 491-546): raw nn.Parameters, activations
(normalize / exp / sigmoid / cat), n_active_features, background().  It exists so that the train-step
harness and the tests drive `Tracer.render` through the same interface `MixtureOfGaussians` would.
"""
import numpy as np
import torch


class GaussianModel(torch.nn.Module):
    def __init__(self, scene: dict, device="cuda", sh_degree=3, background_color="black", max_n_features=3):
        super().__init__()
        t = lambda a: torch.nn.Parameter(torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device))
        feats = np.asarray(scene["features"], np.float32)
        dens = np.clip(np.asarray(scene["density"], np.float64), 1e-6, 1 - 1e-6)
        self.positions = t(scene["positions"])
        self.rotation = t(scene["rotation"])                                  # pre-activation: un-normalised quaternion
        self.scale = t(np.log(np.asarray(scene["scale"], np.float64)))        # pre-activation: log-scale
        self.density = t(np.log(dens / (1 - dens)))                           # pre-activation: logit
        self.features_albedo = t(feats[:, :3])
        # model.py:139-154: features_specular is [N, 3 (max_n_features + 1)^2 - 3] (sh_degree_to_specular_dim); max_n_features is
        # conf.model.progressive_training.max_n_features and must equal render.particle_radiance_sph_degree
        self.features_specular = t(feats[:, 3:3 * (int(max_n_features) + 1) ** 2])
        # the reference's activation callables and their names (model.py:163-167, utils/misc.py:45-50, base_gs.yaml:54-55)
        self.density_activation = torch.sigmoid
        self.scale_activation = torch.exp
        self.rotation_activation = torch.nn.functional.normalize
        self.max_n_features = int(max_n_features)
        self.n_active_features = min(int(sh_degree), self.max_n_features)
        self.background_color = background_color
        self.device = device

    @property
    def num_gaussians(self):
        return self.positions.shape[0]

    def get_rotation(self, preactivation=False):          # model.py:83-93
        return self.rotation if preactivation else self.rotation_activation(self.rotation)

    def get_scale(self, preactivation=False):
        return self.scale if preactivation else self.scale_activation(self.scale)

    def get_density(self, preactivation=False):
        return self.density if preactivation else self.density_activation(self.density)

    def get_features_albedo(self):      # model.py:68-72
        return self.features_albedo

    def get_features_specular(self):
        return self.features_specular

    def get_features(self):
        return torch.cat((self.features_albedo, self.features_specular), dim=1)

    def background(self, T_to_world, rays_d, rgb, opacity, train=False):
        # BackgroundColor.forward (model/background.py:78-93), default colour black
        if self.background_color == "white":
            rgb = rgb + (1.0 - opacity)
        elif self.background_color == "random" and train:
            rgb = rgb + torch.rand_like(rays_d) * (1.0 - opacity)
        return rgb, opacity

    def param_groups(self, extent=1.0):
        """Adam groups with the reference's learning rates (configs/base_gs.yaml:81-109, model.py:518-520)."""
        return [
            dict(params=[self.positions], lr=1.6e-4 * extent, name="positions"),
            dict(params=[self.density], lr=0.05, name="density"),
            dict(params=[self.features_albedo], lr=0.0025, name="features_albedo"),
            dict(params=[self.features_specular], lr=0.000125, name="features_specular"),
            dict(params=[self.rotation], lr=0.001, name="rotation"),
            dict(params=[self.scale], lr=0.005, name="scale"),
        ]
