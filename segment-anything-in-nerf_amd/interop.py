"""Name-based parameter import/export between SAMModel and a plain {name: tensor} map.

Names: prop_table, prop_w{i}; field_table, base_w{i}, head_w{i}; {sam,clipseg}_table{j}, {sam,clipseg}_w{i};
conv{0,1}_{w,b}.  Tables are [L*2^T, F] (level-major rows), layers [out, in] -- the torch layouts of the
reference's HashEncoding.hash_table and nn.Linear.weight, so a torch-path checkpoint maps one to one."""
from __future__ import annotations

from typing import Dict

import torch


def _named_modules(model):
    prop = model.proposal_networks[0].mlp_base
    out = {"prop": (prop.encoding, prop.network), "field": (model.field.mlp_base.encoding, None),
           "base": (None, model.field.mlp_base.network), "head": (None, model.field.mlp_head)}
    return out


@torch.no_grad()
def load_named_params(model, params: Dict[str, torch.Tensor]) -> None:
    dev = model.device

    def put_table(enc, t):
        enc.params.copy_(t.to(dev).reshape(-1))

    def put_mlp(net, prefix):
        ws, i = [], 0
        while f"{prefix}_w{i}" in params:
            ws.append(params[f"{prefix}_w{i}"].to(dev))
            i += 1
        net.load_weights(ws)

    prop = model.proposal_networks[0].mlp_base
    put_table(prop.encoding, params["prop_table"])
    put_mlp(prop.network, "prop")
    put_table(model.field.mlp_base.encoding, params["field_table"])
    put_mlp(model.field.mlp_base.network, "base")
    put_mlp(model.field.mlp_head, "head")
    if getattr(model.config, "distill_sam", False):
        for j, enc in enumerate(model.sam_field.clip_encs):
            put_table(enc, params[f"sam_table{j}"])
        put_mlp(model.sam_field.sam_net, "sam")
        if model.config.use_clipseg_feature:
            for j, enc in enumerate(model.sam_field.clipseg_encs):
                put_table(enc, params[f"clipseg_table{j}"])
            put_mlp(model.sam_field.clipseg_net, "clipseg")
        if "conv0_w" in params:
            model.conv_head[0].weight.copy_(params["conv0_w"].to(dev))
            model.conv_head[0].bias.copy_(params["conv0_b"].to(dev))
            model.conv_head[2].weight.copy_(params["conv1_w"].to(dev))
            model.conv_head[2].bias.copy_(params["conv1_b"].to(dev))


def named_grads(model) -> Dict[str, torch.Tensor]:
    """Gradients under the same names (reads the arena slices, i.e. `param.main_grad`)."""
    g: Dict[str, torch.Tensor] = {}

    def grad_of(p):
        mg = getattr(p, "main_grad", None)
        return mg if mg is not None else p.grad

    def table(enc, name):
        g[name] = grad_of(enc.params).view(-1, enc.n_features_per_level)

    def mlp(net, prefix):
        flat, off = grad_of(net.params), 0
        for i, (o, k) in enumerate(net.layer_shapes):
            g[f"{prefix}_w{i}"] = flat[off:off + o * k].view(o, k)
            off += o * k

    prop = model.proposal_networks[0].mlp_base
    table(prop.encoding, "prop_table")
    mlp(prop.network, "prop")
    table(model.field.mlp_base.encoding, "field_table")
    mlp(model.field.mlp_base.network, "base")
    mlp(model.field.mlp_head, "head")
    if getattr(model.config, "distill_sam", False):
        for j, enc in enumerate(model.sam_field.clip_encs):
            table(enc, f"sam_table{j}")
        mlp(model.sam_field.sam_net, "sam")
        if model.config.use_clipseg_feature:
            for j, enc in enumerate(model.sam_field.clipseg_encs):
                table(enc, f"clipseg_table{j}")
            mlp(model.sam_field.clipseg_net, "clipseg")
        g["conv0_w"], g["conv0_b"] = grad_of(model.conv_head[0].weight), grad_of(model.conv_head[0].bias)
        g["conv1_w"], g["conv1_b"] = grad_of(model.conv_head[2].weight), grad_of(model.conv_head[2].bias)
    return g
