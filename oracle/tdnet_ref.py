"""oracle/tdnet_ref.py -- TEST INFRASTRUCTURE ONLY (CPU restatement of the TDNet hot path).

A functional PyTorch-CPU restatement of the reference's Transformer-based Deformation Network,
following the reference's own op sequence tensor for tensor (each function cites the reference
file:line it restates).  It takes a flat ``state_dict`` with the reference's key names, so the very
same procedural weights can be loaded into the imported reference (``oracle/make_golden.py``, only
in the build container), into this oracle and into the HIP product (``nsdp_amd``).

Pinning: this file is checked against outputs of the *imported reference itself* (fixtures under
``tests/golden/`` written by ``oracle/make_golden.py``; see tests/test_oracle_golden.py).  The only
piece the reference cannot execute on a CPU is farthest-point sampling (CUDA-only,
sampling.cpp:82-84); that is supplied by the literal kernel emulation in oracle/pointnet2_ref.c.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import pointnet2_ref

# /root/reference/config/deform4d/forward.yaml:23-41 (all 13 shipped YAMLs share this block)
DEFAULT_MODEL_CFG = {
    "type": "forward",
    "use_normals": False,
    "encoder": "pointransformer",
    "encoder_kwargs": {
        "npoints_per_layer": [5000, 500, 100],
        "nneighbor": 16,
        "nneighbor_reduced": 10,
        "nfinal_transformers": 3,
        "d_transformer": 256,
        "d_reduced": 120,
        "full_SA": True,
    },
    "decoder": "crossatten",
    "decoder_kwargs": {"dim_inp": 256, "dim": 200, "nneigh": 7, "hidden_dim": 128, "out_dim": 3},
}


# --------------------------------------------------------------------------------------------
# L1 primitives
# --------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """model/utils.py:39-55."""
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def index_points(points, idx):
    """model/utils.py:58-70."""
    raw_size = idx.size()
    idx = idx.reshape(raw_size[0], -1)
    res = torch.gather(points, 1, idx[..., None].expand(-1, -1, points.size(-1)))
    return res.reshape(*raw_size, -1)


def furthest_point_sample(xyz, npoint):
    """pointnet2_utils.py:34-65 -> sampling.cpp:66-87 -> sampling_gpu.cu:69-173 (CPU emulation)."""
    idx = pointnet2_ref.furthest_point_sampling(xyz.detach().cpu().numpy(), int(npoint))
    return torch.from_numpy(idx)


def knn_indices(query, source, k):
    """square_distance(...).argsort()[:, :, :k] (encoder/blocks.py:101-102, :287-288;
    decoder/blocks.py:50-52)."""
    with torch.no_grad():
        return square_distance(query, source).argsort()[:, :, :k]


def compute_l2_error(points_pred, points_gt):
    """model/utils.py:8-11."""
    loss = torch.mean(torch.sub(points_pred, points_gt).pow(2).sum(dim=2) / 2.0)
    return loss.sum(-1).mean()


# --------------------------------------------------------------------------------------------
# layer helpers working on a flat state_dict
# --------------------------------------------------------------------------------------------
class _SD:
    def __init__(self, sd, prefix, training, tape=None):
        self.sd, self.prefix, self.training, self.tape = sd, prefix, training, tape

    def sub(self, name):
        return _SD(self.sd, self.prefix + name + ".", self.training, self.tape)

    def get(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd

    def linear(self, name, x):
        b = self.get(name + ".bias") if self.has(name + ".bias") else None
        return F.linear(x, self.get(name + ".weight"), b)

    def conv1x1(self, name, x_bcn):
        return F.conv1d(x_bcn, self.get(name + ".weight"), self.get(name + ".bias"))

    def bn(self, name, x_bcn):
        """nn.BatchNorm1d on (B,C,n): batch statistics when training (momentum 0.1, eps 1e-5)."""
        nbt = self.prefix + name + ".num_batches_tracked"
        if self.training and nbt in self.sd:
            self.sd[nbt] += 1
        return F.batch_norm(
            x_bcn, self.get(name + ".running_mean"), self.get(name + ".running_var"),
            self.get(name + ".weight"), self.get(name + ".bias"), self.training, 0.1, 1e-5)

    def mlp2(self, name, x):
        """nn.Sequential(Linear, ReLU, Linear) -- fc_delta / fc_gamma / fc_middle."""
        return self.linear(name + ".2", F.relu(self.linear(name + ".0", x)))

    def record(self, key, value):
        if self.tape is not None:
            self.tape[self.prefix + key] = value.detach().clone()


def transformer_block(m: _SD, xyz, feats, k, pos_only=False, group_all=False):
    """TransformerBlock.forward, model/encoder/blocks.py:86-134."""
    with torch.no_grad():
        if group_all:
            b, n, _ = xyz.shape
            knn_idx = torch.arange(n).unsqueeze(0).unsqueeze(1).repeat(b, n, 1)
        else:
            knn_idx = knn_indices(xyz, xyz, k)
    m.record("knn_idx", knn_idx)
    knn_xyz = index_points(xyz, knn_idx)
    if not pos_only:
        q_attn = m.linear("w_qs", feats)
        k_attn = index_points(m.linear("w_ks", feats), knn_idx)
        v_attn = index_points(m.linear("w_vs", feats), knn_idx)
    pos_encode = m.mlp2("fc_delta", xyz[:, :, None] - knn_xyz)
    if not pos_only:
        attn = m.mlp2("fc_gamma", q_attn[:, :, None] - k_attn + pos_encode)
    else:
        attn = m.mlp2("fc_gamma", pos_encode)
    attn = F.softmax(attn, dim=-2)
    if not pos_only:
        res = torch.einsum("bmnf,bmnf->bmf", attn, v_attn + pos_encode)
        res = res + feats
    else:
        res = torch.einsum("bmnf,bmnf->bmf", attn, pos_encode)
    res = m.bn("bn", res.permute(0, 2, 1)).permute(0, 2, 1)
    m.record("out", res)
    return res


def elementwise_mlp(m: _SD, x):
    """ElementwiseMLP.forward, model/encoder/blocks.py:153-159."""
    x = x.permute(0, 2, 1)
    y = m.bn("bn3", x + F.relu(m.bn("bn2", m.conv1x1("conv2", F.relu(m.bn("bn1", m.conv1x1("conv1", x)))))))
    y = y.permute(0, 2, 1)
    m.record("out", y)
    return y


def transformer_set_abstraction(m: _SD, xyz, points, npoint, nneigh):
    """TransformerSetAbstraction.forward, model/encoder/blocks.py:270-314."""
    B, N, C = xyz.shape
    with torch.no_grad():
        fps_idx = furthest_point_sample(xyz, npoint)
        new_xyz = index_points(xyz, fps_idx.long())
        idx = knn_indices(new_xyz, xyz, nneigh)
    m.record("fps_idx", fps_idx)
    m.record("knn_idx", idx)
    q_attn = index_points(m.linear("w_qs", points), fps_idx.long())
    k_attn = index_points(m.linear("w_ks", points), idx)
    v_attn = index_points(m.linear("w_vs", points), idx)
    grouped_xyz = index_points(xyz, idx)
    pos_encode = m.mlp2("fc_delta1", grouped_xyz - new_xyz.view(B, npoint, 1, C))
    attn = m.mlp2("fc_gamma1", q_attn[:, :, None] - k_attn + pos_encode)
    attn = F.softmax(attn, dim=-2)
    res1 = torch.einsum("bmnf,bmnf->bmf", attn, v_attn + pos_encode)
    res1 = res1 + m.conv1x1("conv2", F.relu(m.bn("bn1", m.conv1x1("conv1", res1.permute(0, 2, 1))))).permute(0, 2, 1)
    res1 = m.bn("bnorm0", res1.permute(0, 2, 1)).permute(0, 2, 1)
    q_attn = m.linear("w_qs2", res1)
    k_attn = index_points(m.linear("w_ks2", points), idx)
    v_attn = index_points(m.linear("w_vs2", points), idx)
    attn = m.mlp2("fc_gamma2", q_attn[:, :, None] - k_attn + pos_encode)
    attn = F.softmax(attn, dim=-2)
    res2 = torch.einsum("bmnf,bmnf->bmf", attn, v_attn + pos_encode)
    new_points = m.bn("bnorm1", (res1 + res2).permute(0, 2, 1)).permute(0, 2, 1)
    new_points = new_points + index_points(points, fps_idx.long())
    new_points = m.bn("bnorm2", new_points.permute(0, 2, 1)).permute(0, 2, 1)
    m.record("out", new_points)
    return new_xyz, new_points


def point_transformer_encoder(m: _SD, x, kw, has_features):
    """PointTransformerEncoder.forward, model/encoder/pointransformer.py:87-140."""
    npl = kw["npoints_per_layer"]
    nneighbor, nred = kw["nneighbor"], kw["nneighbor_reduced"]
    d_t, d_r = kw["d_transformer"], kw["d_reduced"]
    if has_features:
        feats = m.linear("enc_sdf", x[:, :, 3:])
        xyz = x[:, :, :3].contiguous()
        feats = transformer_block(m.sub("transformer_begin"), xyz, feats, nred)
    else:
        xyz = x
        feats = transformer_block(m.sub("transformer_begin"), xyz, None, nred, pos_only=True)
    for i in range(len(npl) - 1):
        old_n, new_n = npl[i], npl[i + 1]
        xyz, feats = transformer_set_abstraction(
            m.sub(f"transition_downs.{i}.sa"), xyz, feats, new_n, min(nneighbor, old_n))
        feats = elementwise_mlp(m.sub(f"elementwise_extras.{i}"), feats)
        feats = transformer_block(m.sub(f"transformer_downs.{i}"), xyz, feats, min(nneighbor, new_n))
        if i == 0 and d_r != d_t:
            feats = m.linear("fc1", feats)
        feats = elementwise_mlp(m.sub(f"elementwise.{i}"), feats)
    for i in range(kw["nfinal_transformers"]):
        feats = transformer_block(m.sub(f"final_transformers.{i}"), xyz, feats, 2 * nneighbor,
                                  group_all=kw["full_SA"])
        feats = elementwise_mlp(m.sub(f"final_elementwise.{i}"), feats)
    lat_vec = feats.max(dim=1)[0]
    z = m.mlp2("fc_middle", lat_vec)
    m.record("z", z)
    return {"z": z, "anchors": xyz, "anchor_feats": feats}


def pointnet_set_abstraction(m: _SD, xyz, points, npoint, nneigh):
    """PointNetSetAbstraction.forward, model/encoder/blocks.py:184-217 (registry alternate, max-pool SA)."""
    with torch.no_grad():
        fps_idx = furthest_point_sample(xyz, npoint).long()
    new_xyz = index_points(xyz, fps_idx)
    points = m.linear("fc1", points)
    points_ori = index_points(points, fps_idx)
    pt = points.permute(0, 2, 1)
    pt = pt + F.relu(m.bn("bn2", m.conv1x1("conv2", F.relu(m.bn("bn1", m.conv1x1("conv1", pt))))))
    points = pt.permute(0, 2, 1)
    idx = knn_indices(new_xyz, xyz, nneigh)
    new_points = points_ori + torch.max(index_points(points, idx), 2)[0]
    new_points = m.bn("bn", new_points.permute(0, 2, 1)).permute(0, 2, 1)
    return new_xyz, new_points


def pointnetpp_encoder(m: _SD, x, kw, has_features):
    """PointNetPlusPlusEncoder.forward, model/encoder/pointnetplusplus.py:70-96."""
    npl = kw["npoints_per_layer"]
    if has_features:
        feats = m.mlp2("fc_begin", x[:, :, 3:].contiguous())
        xyz = x[:, :, 0:3].contiguous()
    else:
        feats, xyz = m.mlp2("fc_begin", x), x
    for i in range(len(npl) - 1):
        xyz, feats = pointnet_set_abstraction(m.sub(f"transition_downs.{i}.sa"), xyz, feats, npl[i + 1],
                                              min(kw["nneighbor"], npl[i]))
        feats = elementwise_mlp(m.sub(f"elementwise.{i}"), feats)
    for i in range(kw["nfinal_transformers"]):
        feats = transformer_block(m.sub(f"final_transformers.{i}"), xyz, feats, -1, group_all=True)
        feats = elementwise_mlp(m.sub(f"final_elementwise.{i}"), feats)
    return {"z": m.mlp2("fc_middle", feats.max(dim=1)[0]), "anchors": xyz, "anchor_feats": feats}


def point_interp_decoder(m: _SD, xyz_q, enc, kw, n_blocks=5):
    """PointInterpDecoder.forward, model/decoder/interpolation_decoder.py:47-88 (Gaussian kernel, var = 0.2^2)."""
    p, fea = enc["anchors"], enc["anchor_feats"]
    dist = -((p.unsqueeze(1).expand(-1, xyz_q.size(1), -1, -1) - xyz_q.unsqueeze(2)).norm(dim=3) + 10e-6) ** 2
    weight = (dist / 0.2 ** 2).exp()
    weight = weight / weight.sum(dim=2).unsqueeze(-1)
    lat = m.linear("fc0", weight @ fea)
    net = m.linear("fc1", F.relu(lat))
    for i in range(n_blocks):
        net = net + m.linear(f"fc_c.{i}", lat)
        net = net + m.linear(f"blocks.{i}.fc_1", F.relu(m.linear(f"blocks.{i}.fc_0", F.relu(net))))
    return m.linear("fc_out", F.relu(net))


def cross_transformer_block(m: _SD, xyz_q, lat_rep, xyz, points, nneigh, dim):
    """CrossTransformerBlock.forward, model/decoder/blocks.py:48-95 (separate_delta=True)."""
    knn_idx = knn_indices(xyz_q, xyz, nneigh)
    m.record("knn_idx", knn_idx)
    b, nQ, _ = xyz_q.shape
    q_attn = m.linear("w_qs", lat_rep).unsqueeze(1).repeat(1, nQ, 1)
    k_global = m.linear("w_k_global", lat_rep).unsqueeze(1).repeat(1, nQ, 1).unsqueeze(2)
    v_global = m.linear("w_v_global", lat_rep).unsqueeze(1).repeat(1, nQ, 1).unsqueeze(2)
    k_attn = torch.cat([index_points(m.linear("w_ks", points), knn_idx), k_global], dim=2)
    v_attn = torch.cat([index_points(m.linear("w_vs", points), knn_idx), v_global], dim=2)
    xyz = index_points(xyz, knn_idx)
    pos_encode = m.mlp2("fc_delta", xyz_q[:, :, None] - xyz)
    pos_encode = torch.cat([pos_encode, torch.zeros([b, nQ, 1, dim])], dim=2)
    pos_encode2 = m.mlp2("fc_delta", xyz_q[:, :, None] - xyz)  # evaluated twice, :81-84
    pos_encode2 = torch.cat([pos_encode2, torch.zeros([b, nQ, 1, dim])], dim=2)
    attn = m.mlp2("fc_gamma", q_attn[:, :, None] - k_attn + pos_encode)
    attn = F.softmax(attn, dim=-2)
    res = torch.einsum("bmnf,bmnf->bmf", attn, v_attn + pos_encode2)
    m.record("out", res)
    return res


def cross_transformer_decoder(m: _SD, xyz_q, enc, kw, n_blocks=5):
    """CrossTransformerDecoder.forward, model/decoder/crosstransformer_decoder.py:45-70."""
    lat = cross_transformer_block(m.sub("ct1"), xyz_q, enc["z"], enc["anchors"], enc["anchor_feats"],
                                  kw["nneigh"], kw["dim"])
    net = m.linear("init_enc", lat)
    for i in range(n_blocks):
        net = net + m.linear(f"fc_c.{i}", lat)
        # ResnetBlockFC.forward, model/decoder/blocks.py:133-142 (size_in == size_out: no shortcut)
        h = m.linear(f"blocks.{i}.fc_0", F.relu(net))
        dx = m.linear(f"blocks.{i}.fc_1", F.relu(h))
        net = net + dx
    return m.linear("fc_out", F.relu(net))


def deformation_network(sd, model_cfg, points, surface_samples_inputs, no_input_corr, training,
                        prefix="", tape=None):
    """Deformation_Networks.forward, model/deformation_networks.py:43-60 (+ ctor logic :17-30)."""
    m = _SD(sd, prefix, training, tape)
    assert not model_cfg.get("use_normals", False)
    encoder = {"pointransformer": point_transformer_encoder, "pointnet++": pointnetpp_encoder}[model_cfg["encoder"]]
    if no_input_corr:
        enc = encoder(m.sub("encoder"), surface_samples_inputs[:, :, 0:3].contiguous(),
                      model_cfg["encoder_kwargs"], has_features=False)
    else:
        enc = encoder(m.sub("encoder"), surface_samples_inputs, model_cfg["encoder_kwargs"], has_features=True)
    if tape is not None:
        tape[prefix + "anchors"] = enc["anchors"].detach().clone()
        tape[prefix + "anchor_feats"] = enc["anchor_feats"].detach().clone()
    if model_cfg["decoder"] == "interp":
        return point_interp_decoder(m.sub("decoder"), points, enc, model_cfg["decoder_kwargs"])
    return cross_transformer_decoder(m.sub("decoder"), points, enc, model_cfg["decoder_kwargs"])


def model_forward(sd, model_cfg, data, training=False, tape=None, queries_key="space_samples_src"):
    """build_model dispatch (model/__init__.py:43-118): 'forward' | 'backward' | 'arbitrary'."""
    mtype = model_cfg["type"]
    inputs = data["surface_samples_inputs"]
    q = data[queries_key]
    if mtype == "forward":
        return deformation_network(sd, model_cfg, q, inputs, False, training, "", tape)
    if mtype == "backward":
        return deformation_network(sd, model_cfg, q, inputs, True, training, "", tape)
    if mtype == "arbitrary":
        # FlowArbitrary.forward, model/flow_arbitrary.py:15-27 (+ input split :33-36)
        src, tgt, mask = inputs[:, :, 0:3], inputs[:, :, 3:6], inputs[:, :, 6:7]
        # taps of the canonicalize net come from its LAST call (forward hooks overwrite), i.e. s2c
        q2c = deformation_network(sd, model_cfg, q, src, True, training, "model_canonicalize.", None)
        s2c = deformation_network(sd, model_cfg, src, src, True, training, "model_canonicalize.", tape)
        return deformation_network(sd, model_cfg, q2c, torch.cat([s2c, tgt, mask], dim=-1).contiguous(),
                                   False, training, "model_deform.", tape)
    raise NotImplementedError(mtype)


def trainable(sd):
    """Names of the entries that are nn.Parameters in the reference (everything but BN buffers)."""
    return [k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def to_torch_state(np_state, requires_grad=False):
    sd = {}
    for k, v in np_state.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if requires_grad and t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        sd[k] = t
    return sd


def train_step(sd, model_cfg, data, optimizer):
    """train_on_batch_with_cano / _with_arbitrary (deformation_networks.py:63-77,
    flow_arbitrary.py:30-48): zero_grad, forward, l2 loss, backward, optimizer.step, loss.item()."""
    optimizer.zero_grad()
    pred = model_forward(sd, model_cfg, data, training=True)
    loss = compute_l2_error(pred, data["space_samples_tgt"])
    loss.backward()
    optimizer.step()
    return loss.item()
