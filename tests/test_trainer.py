"""`Trainer` is used UNCHANGED: a few optimisation steps on the accelerated model through the
reference training loop (src/transformers/trainer.py:1895-1963 training_step -> compute_loss -> model(**inputs) ->
accelerator.backward), here on the CPU execution model of the kernels; plus gradient checkpointing
(modeling_layers.py:79-114 re-runs our layer forward inside backward)."""
import copy

import pytest
import torch

import transformers_amd
from transformers import LlamaConfig, LlamaForCausalLM, Trainer, TrainingArguments


class Toy(torch.utils.data.Dataset):
    def __init__(self, vocab, n=16, s=32):
        g = torch.Generator().manual_seed(0)
        self.x = torch.randint(0, vocab, (n, s), generator=g)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return {"input_ids": self.x[i], "labels": self.x[i]}


def _model():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=64,
                      rms_norm_eps=1e-5, attn_implementation="eager")
    return LlamaForCausalLM(cfg).bfloat16()


@pytest.mark.timeout(600)
def test_trainer_runs_unchanged_on_accelerated_model(tmp_path):
    from emu_backend import emu_backend

    with emu_backend():
        model = transformers_amd.accelerate(_model())
        before = copy.deepcopy(model.state_dict())
        args = TrainingArguments(output_dir=str(tmp_path), max_steps=3, per_device_train_batch_size=4,
                                 learning_rate=1e-2, report_to=[], use_cpu=True, bf16=False, save_strategy="no",
                                 logging_steps=1, dataloader_pin_memory=False, disable_tqdm=True)
        tr = Trainer(model=model, args=args, train_dataset=Toy(256))
        out = tr.train()
    assert out.global_step == 3 and torch.isfinite(torch.tensor(out.training_loss))
    changed = [k for k, v in model.state_dict().items() if not torch.equal(v, before[k])]
    assert len(changed) >= len(before) - 2  # every trainable tensor moved
    # fused QKV / gate|up buffers followed the optimizer (the parameters are views of them)
    att = model.model.layers[0].self_attn
    assert att._fused()._coherent()
    assert torch.equal(att._fused().weight()[: att.q_proj.weight.shape[0]], att.q_proj.weight)


@pytest.mark.timeout(600)
def test_trainer_with_fused_adamw(tmp_path):
    """`Trainer(optimizers=(TamdAdamW(...), None))`: the reference loop drives our fused optimizer step (section 8 f2);
    the fused QKV / gate|up buffers are updated through the per-parameter views."""
    from emu_backend import emu_backend

    with emu_backend():
        model = transformers_amd.accelerate(_model())
        before = copy.deepcopy(model.state_dict())
        opt = transformers_amd.TamdAdamW(model.parameters(), lr=1e-2, weight_decay=0.01)
        args = TrainingArguments(output_dir=str(tmp_path), max_steps=2, per_device_train_batch_size=4,
                                 report_to=[], use_cpu=True, bf16=False, save_strategy="no", logging_steps=1,
                                 dataloader_pin_memory=False, disable_tqdm=True, lr_scheduler_type="constant")
        out = Trainer(model=model, args=args, train_dataset=Toy(256), optimizers=(opt, None)).train()
    assert out.global_step == 2 and torch.isfinite(torch.tensor(out.training_loss))
    changed = [k for k, v in model.state_dict().items() if not torch.equal(v, before[k])]
    assert len(changed) >= len(before) - 2
    att = model.model.layers[0].self_attn
    assert att._fused()._coherent()
    assert all(s["exp_avg"].dtype == torch.bfloat16 for s in opt.state.values())


def test_gradient_checkpointing_recomputes_through_fused_layer():
    from emu_backend import emu_backend

    with emu_backend():
        a = transformers_amd.accelerate(_model()).train()
        b = transformers_amd.accelerate(_model()).train()
        b.gradient_checkpointing_enable()
        ids = torch.randint(0, 256, (2, 24))
        la = a(input_ids=ids, labels=ids, use_cache=False).loss
        lb = b(input_ids=ids, labels=ids, use_cache=False).loss
        la.backward()
        lb.backward()
    assert torch.equal(la, lb)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.grad, q.grad), n  # pure ops: recomputation is bit-identical
