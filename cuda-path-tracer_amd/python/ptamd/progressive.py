"""Progressive, resumable accumulation on top of pt_render_accumulate.

The reference renders a fixed sample count in one go and keeps nothing (src/renderer.cu:233-262).
Here the device-resident split buffers (src/renderer.cu:194-197) are the state of a render: samples
are added in any number of calls, each call covering the next sample indices, and a checkpoint is
the split buffers plus the sample counter.  Sample s of pixel p always draws from the Philox stream
(seed; p, s), so a resumed render adds exactly the samples an uninterrupted one would have.
"""
import os

import numpy as np

from . import DeviceBuffer, check, lib


class Accumulator:
    def __init__(self, scene, cam, seed=1337, n_split=4, device=0, first_sample=0, tag=""):
        """first_sample: index of the first sample this accumulator renders (a rank of a sharded render starts
        at its shard); tag: free text stored in checkpoints and compared on load (e.g. a hash of the scene)."""
        self.scene, self.cam, self.seed, self.n_split, self.device = scene, cam, int(seed), int(n_split), device
        self.first_sample, self.tag = int(first_sample), str(tag)
        self.n_pixels = cam.res_x * cam.res_y
        self.split = DeviceBuffer(device, self.n_split * self.n_pixels * 12)
        self.split.zero()
        self.samples_done = 0

    def add(self, n_samples, stream=None):
        """Adds the next n_samples samples of every pixel: indices first_sample + [samples_done, samples_done + n_samples)."""
        if n_samples <= 0:
            return
        b = self.first_sample + self.samples_done
        self.scene.render_accumulate(self.cam, self.seed, b, b + int(n_samples), self.n_split, self.split.ptr, False, stream)
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
        """Checkpoint: split buffers + counters (numpy .npz, no pickling), written to exactly `path` (no suffix is
        appended) through a temporary file and an atomic rename, so that an interrupted write never destroys the
        previous checkpoint."""
        split = self.split.download((self.n_split, self.cam.res_y, self.cam.res_x, 3))
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "wb") as f:
            np.savez(f, split=split, samples_done=np.int64(self.samples_done), seed=np.int64(self.seed),
                     first_sample=np.int64(self.first_sample), res=np.array([self.cam.res_x, self.cam.res_y], np.int64),
                     tag=np.frombuffer(self.tag.encode("utf-8"), dtype=np.uint8))
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)

    def load(self, path, max_samples=None):
        """Restores a checkpoint written by save(); rejects one made with another camera resolution, split count,
        seed, first sample or tag, or one that already holds more than max_samples samples."""
        with np.load(path, allow_pickle=False) as z:
            if tuple(z["res"]) != (self.cam.res_x, self.cam.res_y) or z["split"].shape[0] != self.n_split:
                raise ValueError("checkpoint does not match this camera / split count")
            if int(z["seed"]) != self.seed:
                raise ValueError("checkpoint was rendered with another seed")
            if "first_sample" in z and int(z["first_sample"]) != self.first_sample:
                raise ValueError("checkpoint starts at another sample index")
            tag = bytes(z["tag"]).decode("utf-8") if "tag" in z else ""
            if tag != self.tag:
                raise ValueError("checkpoint belongs to another scene or render setting")
            done = int(z["samples_done"])
            if done < 0 or (max_samples is not None and done > max_samples):
                raise ValueError(f"checkpoint holds {done} samples per pixel, more than the {max_samples} this render asks for")
            self.split.upload(np.ascontiguousarray(z["split"], dtype=np.float32))
            self.samples_done = done

    def close(self):
        self.split.free()
