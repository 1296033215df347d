"""
CPU oracle for the BertForDiffusion noise predictor (TEST INFRASTRUCTURE ONLY).

Restates, op for op, the eval-mode forward of
``foldingdiff/modelling.py:384-484`` (BertForDiffusionBase.forward) and the
third-party ``transformers==4.11.3`` BertEncoder it calls
(``modelling.py:271`` constructs it, ``:473-480`` calls it).  Module / parameter
names follow ``modelling.py:239-295`` and HF BERT so that a real Lightning
``.ckpt`` ``state_dict`` loads with ``strict=True``.

Works in float32 (the reference dtype) or float64 (``.double()``) -- the fp64
instance is the "truth" used to measure rounding noise of both the fp32 oracle
and the HIP kernels.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn


class OracleConfig:
    """The subset of HF ``BertConfig`` the hot path reads
    (``bin/train.py:425-435`` builds it; ``config.json`` persists it)."""

    def __init__(
        self,
        hidden_size: int = 384,
        num_attention_heads: int = 12,
        intermediate_size: int = 768,
        num_hidden_layers: int = 12,
        max_position_embeddings: int = 128,
        position_embedding_type: str = "relative_key",
        layer_norm_eps: float = 1e-12,
        hidden_act: str = "gelu",
        initializer_range: float = 0.02,
        **_ignored,
    ):
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.max_position_embeddings = max_position_embeddings
        self.position_embedding_type = position_embedding_type
        self.layer_norm_eps = layer_norm_eps
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        assert hidden_act == "gelu", "reference configs use exact-erf gelu"


# --------------------------------------------------------------------------
# modelling.py:42-71
class GaussianFourierProjection(nn.Module):
    def __init__(self, embed_dim: int, scale: float = 2 * math.pi):
        super().__init__()
        self.register_buffer("W", torch.randn(embed_dim // 2) * scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim > 1:
            x = x.squeeze()
        elif x.ndim < 1:
            x = x.unsqueeze(0)
        # modelling.py:69 -- evaluated left to right: ((x*W)*2)*pi, in W's dtype
        x_proj = x[:, None] * self.W[None, :] * 2 * torch.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


# modelling.py:74-93
class SinusoidalPositionEmbeddings(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, time: torch.Tensor) -> torch.Tensor:
        half_dim = self.dim // 2
        e = math.log(10000) / (half_dim - 1)
        e = torch.exp(torch.arange(half_dim, device=time.device) * -e)
        e = time[:, None] * e[None, :]
        return torch.cat((e.sin(), e.cos()), dim=-1)


# modelling.py:132-170
class BertEmbeddings(nn.Module):
    def __init__(self, config: OracleConfig):
        super().__init__()
        self.position_embedding_type = config.position_embedding_type
        if self.position_embedding_type == "absolute":
            self.position_embeddings = nn.Embedding(
                config.max_position_embeddings, config.hidden_size
            )
            self.register_buffer(
                "position_ids",
                torch.arange(config.max_position_embeddings).expand((1, -1)),
            )
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, input_embeds, position_ids):
        e = input_embeds
        if self.position_embedding_type == "absolute":
            e = e + self.position_embeddings(position_ids)
        return self.LayerNorm(e)  # dropout = identity in eval


# --- HF transformers 4.11.3 BertSelfAttention (third party; restated) -----
class BertSelfAttention(nn.Module):
    def __init__(self, config: OracleConfig):
        super().__init__()
        d, h = config.hidden_size, config.num_attention_heads
        assert d % h == 0
        self.num_attention_heads = h
        self.attention_head_size = d // h
        self.query = nn.Linear(d, d)
        self.key = nn.Linear(d, d)
        self.value = nn.Linear(d, d)
        self.position_embedding_type = config.position_embedding_type
        if self.position_embedding_type in ("relative_key", "relative_key_query"):
            self.max_position_embeddings = config.max_position_embeddings
            self.distance_embedding = nn.Embedding(
                2 * config.max_position_embeddings - 1, self.attention_head_size
            )

    def _split(self, x):
        b, l, _ = x.shape
        return x.view(b, l, self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, ext_mask):
        q = self._split(self.query(hidden_states))
        k = self._split(self.key(hidden_states))
        v = self._split(self.value(hidden_states))
        scores = torch.matmul(q, k.transpose(-1, -2))
        if self.position_embedding_type in ("relative_key", "relative_key_query"):
            L = hidden_states.shape[1]
            pos_l = torch.arange(L, dtype=torch.long).view(-1, 1)
            pos_r = torch.arange(L, dtype=torch.long).view(1, -1)
            distance = pos_l - pos_r
            pe = self.distance_embedding(distance + self.max_position_embeddings - 1)
            pe = pe.to(dtype=q.dtype)
            if self.position_embedding_type == "relative_key":
                scores = scores + torch.einsum("bhld,lrd->bhlr", q, pe)
            else:
                scores = (
                    scores
                    + torch.einsum("bhld,lrd->bhlr", q, pe)
                    + torch.einsum("bhrd,lrd->bhlr", k, pe)
                )
        scores = scores / math.sqrt(self.attention_head_size)
        scores = scores + ext_mask
        probs = torch.softmax(scores, dim=-1)  # dropout = identity in eval
        ctx = torch.matmul(probs, v)
        ctx = ctx.permute(0, 2, 1, 3).contiguous()
        return ctx.view(ctx.shape[0], ctx.shape[1], -1)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dense(hidden_states) + input_tensor)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, h, ext_mask):
        return self.output(self.self(h, ext_mask), h)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, h):
        return torch.nn.functional.gelu(self.dense(h))  # exact erf gelu


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dense(hidden_states) + input_tensor)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, h, ext_mask):
        a = self.attention(h, ext_mask)
        return self.output(self.intermediate(a), a)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, h, ext_mask):
        for layer in self.layer:
            h = layer(h, ext_mask)
        return h


# modelling.py:173-208
class AnglesPredictor(nn.Module):
    def __init__(self, d_model: int, d_out: int, eps: float = 1e-12):
        super().__init__()
        self.dense1 = nn.Linear(d_model, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)
        self.dense2 = nn.Linear(d_model, d_out)

    def forward(self, x):
        return self.dense2(self.layer_norm(torch.nn.functional.gelu(self.dense1(x))))


class OracleBertForDiffusion(nn.Module):
    """Eval-mode restatement of ``BertForDiffusionBase`` (modelling.py:211-484)."""

    def __init__(
        self,
        config: OracleConfig,
        ft_is_angular: Sequence[bool] = (True,) * 6,
        time_encoding: str = "gaussian_fourier",
        decoder: str = "mlp",
    ):
        super().__init__()
        self.config = config
        self.ft_is_angular = list(ft_is_angular)
        self.n_inputs = len(self.ft_is_angular)
        self.inputs_to_hidden_dim = nn.Linear(self.n_inputs, config.hidden_size)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        if decoder == "linear":
            self.token_decoder = nn.Linear(config.hidden_size, self.n_inputs)
        elif decoder == "mlp":
            self.token_decoder = AnglesPredictor(config.hidden_size, self.n_inputs)
        else:
            raise ValueError(f"Unrecognized decoder: {decoder}")
        if time_encoding == "gaussian_fourier":
            self.time_embed = GaussianFourierProjection(config.hidden_size)
        elif time_encoding == "sinusoidal":
            self.time_embed = SinusoidalPositionEmbeddings(config.hidden_size)
        else:
            raise ValueError(f"Unknown time encoding: {time_encoding}")
        self.time_encoding = time_encoding
        self.decoder = decoder
        self.init_weights_hf()
        self.eval()
        # Optional override: a fixed [T, d] time-embedding table (float32 values
        # computed with the reference's fp32 op order).  Used when running the
        # oracle in float64 so that the ~1e5-rad sin/cos arguments do not
        # dominate the fp32-vs-fp64 comparison.
        self.time_table: Optional[torch.Tensor] = None

    # HF BertPreTrainedModel._init_weights (4.11.3): Linear/Embedding ~ N(0, range),
    # bias 0, LayerNorm weight 1 / bias 0.
    def init_weights_hf(self):
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=std)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(mean=0.0, std=std)
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)

    def time_encode(self, timestep: torch.Tensor, dtype) -> torch.Tensor:
        if self.time_table is not None:
            return self.time_table[timestep.reshape(-1).long()].to(dtype)
        # In the reference the integer timestep multiplies the float32 buffer W
        # -> float32 arithmetic (modelling.py:69).
        return self.time_embed(timestep.squeeze(dim=-1)).to(dtype)

    @torch.no_grad()
    def forward(self, inputs, timestep, attention_mask, position_ids=None, **_unused):
        b, l = inputs.shape[:2]
        assert attention_mask is not None
        if position_ids is None:
            position_ids = torch.arange(l).expand(b, -1)
        assert attention_mask.dim() == 2
        assert inputs.dim() == 3
        ext = attention_mask[:, None, None, :].to(inputs.dtype)
        ext = (1.0 - ext) * -10000.0  # modelling.py:450-452
        h = self.inputs_to_hidden_dim(inputs)  # :464
        h = self.embeddings(h, position_ids=position_ids.long())  # :467
        te = self.time_encode(timestep, h.dtype).unsqueeze(1)  # :471
        h = h + te  # :472
        h = self.encoder(h, ext)  # :473-480
        return self.token_decoder(h)  # :482-484


def synthetic_model(
    config: OracleConfig,
    ft_is_angular: Sequence[bool] = (True,) * 6,
    time_encoding: str = "gaussian_fourier",
    decoder: str = "mlp",
    seed: int = 0,
    perturb: bool = True,
) -> OracleBertForDiffusion:
    """Seeded synthetic weights of the released architecture.

    HF init (N(0, 0.02), zero bias, LN gamma=1/beta=0) as in SURVEY 8(d).  With
    ``perturb=True`` biases and LayerNorm affine parameters are additionally
    randomised so that parity tests exercise every parameter (an all-zero bias
    would hide a dropped ``+ b``).
    """
    g = torch.Generator().manual_seed(seed)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    try:
        m = OracleBertForDiffusion(config, ft_is_angular, time_encoding, decoder)
    finally:
        torch.set_rng_state(state)
    if perturb:
        for name, p in m.named_parameters():
            if name.endswith("bias"):
                p.data = torch.randn(p.shape, generator=g) * 0.05
            elif "LayerNorm.weight" in name or "layer_norm.weight" in name:
                p.data = 1.0 + torch.randn(p.shape, generator=g) * 0.1
    return m
