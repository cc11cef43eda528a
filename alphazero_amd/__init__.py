"""MI355X-native batched AlphaZero self-play engine (HIP kernels behind a C ABI).

Host-side modules mirror the reference's plug-in contracts (env_path /
state_converter_path / gameplay_path, evaluator, MCTSearch, Player); all
compute runs in alphazero_amd/csrc/libazamd.so.  There is no CPU fallback:
importing the binding without the built library, or creating an engine without
a GPU, raises.
"""
__all__ = ["Engine", "EngineError", "IllegalMove"]


def __getattr__(name):
    if name in __all__:
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
