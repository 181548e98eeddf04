"""The reference's evaluation run at its real size through the drop-in harness: eval_maskbit.py:107-137 -- labels = randperm(1000).repeat(50),
50 000 images in batches of 64 (BASELINE configs[2]: 12-bit, 64 steps, CFG 7.1 cosine), uint8 NHWC on the host -- on synthetic weights.  Checks what a
production run has to hold for hours: steady throughput (first / last tenth), no growth of device memory, finite pixels, saturation counters at 0, every
class covered 50 times.  usage: python tools/soak_eval.py [images = 50000] [batch = 64]   (~25 min of GPU time at the default)"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench                                                   # build_models / SAMPLER / NUM_STEPS of the headline workload


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from maskbit_amd import eval_labels, generate_uint8
    from maskbit_amd.telemetry import ClockSampler
    dev = torch.device("cuda", 0)
    gen, tok = bench.build_models(dev)
    torch.manual_seed(0)
    labels = eval_labels(dev)
    nb = total // B
    S = bench.SAMPLER
    crc, n_img, sums = 0, 0, np.zeros(3)
    marks, mem = [], []
    cls = np.zeros(1000, dtype=np.int64)
    t0 = time.perf_counter()
    with ClockSampler(0) as clocks:
        it = generate_uint8(gen, tok, labels, B, randomize_temperature=S["randomize_temperature"], mask_schedule_strategy=S["mask_schedule_strategy"],
                            num_steps=bench.NUM_STEPS, guidance_scale=S["guidance_scale"], guidance_annealing=S["guidance_annealing"],
                            scale_pow=S["scale_pow"], total_samples=nb * B)
        for i, u8 in enumerate(it):
            assert u8.shape == (B, 256, 256, 3) and u8.dtype == np.uint8
            crc = zlib.crc32(u8.tobytes(), crc)
            sums += u8.reshape(-1, 3).mean(0)
            n_img += B
            cls += np.bincount(labels[B * i: B * (i + 1)].cpu().numpy(), minlength=1000)
            marks.append(time.perf_counter())
            if i % 50 == 0:
                mem.append(torch.cuda.memory_allocated(dev) >> 20)
                print(f"batch {i + 1}/{nb}: {n_img / (marks[-1] - t0):.2f} images/s so far, device memory {mem[-1]} MiB (reserved {torch.cuda.memory_reserved(dev) >> 20})", flush=True)
    dt = marks[-1] - t0
    tenth = max(2, nb // 10)
    first = (tenth - 1) * B / (marks[tenth - 1] - marks[0]); last = (tenth - 1) * B / (marks[-1] - marks[-tenth])
    tele = clocks.summary()
    print(f"{n_img} images in {dt:.1f} s = {n_img / dt:.2f} images/s (first tenth {first:.2f}, last tenth {last:.2f}); "
          f"clock {tele.get('effective_clock_mhz')} MHz, socket {tele.get('socket_power_w')} W")
    print(f"device memory at the checkpoints (MiB): min {min(mem)} max {max(mem)}; mean pixel per channel {np.round(sums / (n_img / B), 2).tolist()}; crc32 of all bytes {crc:08x}")
    print(f"classes drawn: min {int(cls.min())} max {int(cls.max())} per class over {int(cls.sum())} labels; "
          f"saturated fp16 stores: generator {gen.saturation_count()}, decoder {tok.saturation_count()}")


if __name__ == "__main__":
    main()
