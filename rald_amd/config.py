"""Stand-in for the EasyDict the reference's YAML loader produces (main_generation.py:258-262):
attribute AND .get() access, which is all the model constructors use
(models_radar_generation.py:337-361, :382)."""


class Config(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def shipped_generation_config() -> "Config":
    """The `ar_model` section of configs/generation/*_eval.yml (:117-141)."""
    return Config(cond_type="radar", use_radar_enc=True, unfreeze_radar_enc=True,
                  enc_radar_r_dim=8, enc_radar_a_dim=4, enc_radar_e_dim=2, enc_radar_ch=16, enc_hidden_ch=64,
                  input_radar_r_dim=128, input_radar_a_dim=64, input_radar_e_dim=32, radar_token_channel=512)
