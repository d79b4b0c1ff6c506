#!/usr/bin/env python
"""Where does the host time of an end-to-end training step go?  Times, per step of the real reader + DevicePrefetcher + model.fused_step loop:
the prefetcher's __next__ (reader hand-over, trim, pinned staging, H2D enqueue) and the fused_step call (kernel launches), with and without a
device synchronisation per step.  (bench.py --mode e2e gives the throughput; this says which side of the loop bounds it.)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    from clipcap_amd.encoders.config import EncoderConfig
    from clipcap_amd.model import ClipCapModelPrefixOnly, Config, TrainingConfig
    from clipcap_amd.model.gpt2 import GPT2LM
    from clipcap_amd.train.dataloader import DevicePrefetcher, EmbedDataset
    par = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    c = dict(bench.CONFIGS["2"])
    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as tmp:
        sample = bench._write_e2e_dataset(os.path.join(tmp, "ds"), 102400, c["E"])
        tok = bench._BpeTokenizer(sample)
        ds = EmbedDataset(os.path.join(tmp, "ds"), batch_size=c["B"], tokenizer=tok, max_token_length=c["cap"], reader_parallel_pieces=par)
        lm = GPT2LM(n_embd=c["D"], n_layer=c["n_layer"], n_head=c["n_head"], vocab_size=c["V"], n_positions=c["npos"])
        cfg = Config(language_model="gpt2", prefix_length=c["L"], projection_length=c["P"], transformer_layers=c["N"], transformer_attention_heads=c["H"])
        cfg.encoder_config = EncoderConfig(encoder_embedding_size=c["E"])
        cfg.training_config = TrainingConfig()
        model = ClipCapModelPrefixOnly(cfg, language_model=lm).set_precision("bf16").to(dev)
        model.train()
        t_next, t_step, n = 0.0, 0.0, 0
        # split DevicePrefetcher._load into its parts
        from clipcap_amd.train import dataloader as dl
        sub = {"tokens": 0.0, "emb": 0.0, "shapes": set()}
        parts = {"reader": 0.0, "trim": 0.0, "slot_sync": 0.0, "staging": 0.0, "h2d_enqueue": 0.0}

        class Timed(DevicePrefetcher):
            def _load(self):
                a0 = time.perf_counter()
                try:
                    tokens, emb = next(self.it)
                except StopIteration:
                    return None
                a1 = time.perf_counter()
                tokens = dl.trim_padding(tokens)
                a2 = time.perf_counter()
                i = self._n % len(self._slots)
                self._n += 1
                if self._slots[i] is None:
                    self._slots[i] = [None, None, torch.cuda.Event()]
                slot = self._slots[i]
                if slot[0] is not None:
                    slot[2].synchronize()
                a3 = time.perf_counter()
                b0 = time.perf_counter()
                pt = self._staged(slot, 0, tokens)
                b1 = time.perf_counter()
                pe = self._staged(slot, 1, emb)
                b2 = time.perf_counter()
                sub["tokens"] += b1 - b0
                sub["emb"] += b2 - b1
                sub["shapes"].add((tuple(tokens.shape), str(tokens.dtype), tokens.is_contiguous(), tuple(emb.shape), str(emb.dtype), emb.is_contiguous()))
                a4 = time.perf_counter()
                with torch.cuda.stream(self.stream):
                    out = pt.to(self.device, non_blocking=True), pe.to(self.device, non_blocking=True)
                    slot[2].record(self.stream)
                a5 = time.perf_counter()
                for k, v in zip(parts, (a1 - a0, a2 - a1, a3 - a2, a4 - a3, a5 - a4)):
                    parts[k] += v
                return out

        it = Timed(ds, dev)
        t_all0 = None
        while True:
            a = time.perf_counter()
            try:
                batch = next(it)
            except StopIteration:
                break
            b = time.perf_counter()
            model.fused_step(batch, lr=1e-6)
            cdone = time.perf_counter()
            n += 1
            if n == 50:
                torch.cuda.synchronize()
                t_all0 = time.perf_counter()
                t_next = t_step = 0.0
                for k_ in parts:
                    parts[k_] = 0.0
            elif n > 50:
                t_next += b - a
                t_step += cdone - b
            if n == 350:
                break
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_all0
        k = n - 50
        print(f"reader_parallel_pieces {par}: wall {wall / k * 1e3:.2f} ms per step; host time in next(prefetcher) {t_next / k * 1e3:.2f} ms, in fused_step() {t_step / k * 1e3:.2f} ms")
        print("   next(prefetcher) split: " + ", ".join(f"{k_} {v / k * 1e3:.2f} ms" for k_, v in parts.items()))
        print("   staging split: tokens %.2f ms, emb %.2f ms per step (since start); shapes seen: %s" % (sub["tokens"] / n * 1e3, sub["emb"] / n * 1e3, list(sub["shapes"])[:4]))
        ds.close()
