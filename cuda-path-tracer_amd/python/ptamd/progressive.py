"""Progressive, resumable accumulation on top of pt_render_accumulate.

The reference renders a fixed sample count in one go and keeps nothing (src/renderer.cu:233-262).
Here the device-resident split buffers (src/renderer.cu:194-197) are the state of a render: samples
are added in any number of calls, each call covering the next sample indices, and a checkpoint is
the split buffers plus the sample counter.  Sample s of pixel p always draws from the Philox stream
(seed; p, s), so a resumed render adds exactly the samples an uninterrupted one would have.
"""
import ctypes as C

import numpy as np

from . import DeviceBuffer, check, lib


class Accumulator:
    def __init__(self, scene, cam, seed=1337, n_split=4, device=0):
        self.scene, self.cam, self.seed, self.n_split, self.device = scene, cam, int(seed), int(n_split), device
        self.n_pixels = cam.res_x * cam.res_y
        self.split = DeviceBuffer(device, self.n_split * self.n_pixels * 12)
        self.split.zero()
        self.samples_done = 0

    def add(self, n_samples, stream=None):
        """Adds samples [samples_done, samples_done + n_samples) of every pixel."""
        if n_samples <= 0:
            return
        self.scene.render_accumulate(self.cam, self.seed, self.samples_done, self.samples_done + int(n_samples), self.n_split,
                                     self.split.ptr, False, stream)
        self.samples_done += int(n_samples)

    def radiance_sum(self):
        """Un-normalised H x W x 3 radiance sums (host)."""
        out = DeviceBuffer(self.device, self.n_pixels * 12)
        check(lib().pt_compound_buffers(self.device, self.split.ptr, out.ptr, self.n_pixels, self.n_split, None))
        img = out.download((self.cam.res_y, self.cam.res_x, 3))
        out.free()
        return img

    def image(self):
        """The reference's post chain (src/renderer.cu:277-297) applied to the samples so far."""
        if self.samples_done == 0:
            raise ValueError("no samples accumulated")
        d_sum = DeviceBuffer(self.device, self.n_pixels * 12)
        d_out = DeviceBuffer(self.device, self.n_pixels * 12)
        check(lib().pt_compound_buffers(self.device, self.split.ptr, d_sum.ptr, self.n_pixels, self.n_split, None))
        check(lib().pt_post_chain(self.device, d_sum.ptr, d_out.ptr, self.cam.res_x, self.cam.res_y, self.samples_done, None))
        img = d_out.download((self.cam.res_y, self.cam.res_x, 3))
        d_sum.free()
        d_out.free()
        return img

    def save(self, path):
        """Checkpoint: split buffers + counters (numpy .npz, no pickling)."""
        split = self.split.download((self.n_split, self.cam.res_y, self.cam.res_x, 3))
        np.savez(path, split=split, samples_done=np.int64(self.samples_done), seed=np.int64(self.seed),
                 res=np.array([self.cam.res_x, self.cam.res_y], np.int64))

    def load(self, path):
        with np.load(path, allow_pickle=False) as z:
            if tuple(z["res"]) != (self.cam.res_x, self.cam.res_y) or z["split"].shape[0] != self.n_split:
                raise ValueError("checkpoint does not match this camera / split count")
            if int(z["seed"]) != self.seed:
                raise ValueError("checkpoint was rendered with another seed")
            self.split.upload(np.ascontiguousarray(z["split"], dtype=np.float32))
            self.samples_done = int(z["samples_done"])

    def close(self):
        self.split.free()
