"""InfoTransformer / PositionalEncoding parameter containers (reference: model/transformer_v2.py:22-42,
:154-244, :248-279, :345-391, :447-506, :772-868).  Same constructor arguments, parameter names, shapes and
initialisation (xavier_uniform on every >1-D parameter incl. the RPE GRU, :193-196); arithmetic is in the
HIP engine.
"""
import copy
import math

import torch
from torch import nn


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        assert activation == "relu" and not normalize_before
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)


class TransformerDecoderLayer_TP(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        assert activation == "relu" and not normalize_before
        self.d_model = d_model
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)       # constructed, never run (:816-818)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)                                            # never run
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers, self.norm = num_layers, norm


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers, self.norm, self.return_intermediate = num_layers, norm, return_intermediate


class InfoTransformer(nn.Module):
    def __init__(self, d_model=1024, nhead=4, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False,
                 feat_height=16, feat_width=64):
        super().__init__()
        enc = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(enc, num_encoder_layers, nn.LayerNorm(d_model) if normalize_before else None)
        dec = TransformerDecoderLayer_TP(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec)
        # recurrent positional encoding (:177): fed [W, B, H*d] with batch_first=True -> recurs over the batch axis
        self.gru_encoding = nn.GRU(d_model * feat_height, d_model * feat_height // 2, bidirectional=True, batch_first=True)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead, self.dropout_p = d_model, nhead, dropout
        self.feat_size = (feat_height, feat_width)
