"""Host-side mirror of the reference's Python operator interface for the hot path.

Same names, argument meaning and error behaviour as reference ctransformers/llm.py (`Config` :38-70, ctypes
prototypes :117-208, `LLM` :211-689), written against the same 17-symbol C ABI (include/ctransformers_llm.h),
so a user of `ctransformers.LLM` / `AutoModelForCausalLM` can switch by changing the import.  The shared
library behind it is this repo's HIP build (ctransformers_amd/lib/libctransformers.so); any other library that
exports the ABI (e.g. the reference CPU build used as the test oracle) can be passed via `lib=`.

There is no CPU fallback here: if the HIP library is missing, or no MI355X is visible, construction raises.
"""
import ctypes
import logging
import os
import re
import warnings
from ctypes import POINTER, Structure, c_bool, c_char_p, c_float, c_int, c_void_p
from dataclasses import dataclass, fields
from functools import partial
from pathlib import Path
from typing import Generator, List, Optional, Sequence, Union

logger = logging.getLogger("ctransformers_amd")

c_int_p = POINTER(c_int)
c_float_p = POINTER(c_float)


@dataclass
class Config:
    """Mirror of reference llm.py:38-70 (same fields, same defaults)."""
    # sample
    top_k: int = 40
    top_p: float = 0.95
    temperature: float = 0.8
    repetition_penalty: float = 1.1
    last_n_tokens: int = 64
    seed: int = -1
    # eval
    batch_size: int = 8
    threads: int = -1
    # generate
    max_new_tokens: int = 256
    stop: Optional[Sequence[str]] = None
    stream: bool = False
    reset: bool = True
    # model
    context_length: int = -1
    gpu_layers: int = 0
    mmap: bool = True
    mlock: bool = False

    def to_struct(self):
        return ConfigStruct(context_length=self.context_length, gpu_layers=self.gpu_layers, mmap=self.mmap,
                            mlock=self.mlock)


class ConfigStruct(Structure):
    """`struct Config` passed BY VALUE to ctransformers_llm_create (reference models/llm.h:6-11)."""
    _fields_ = [("context_length", c_int), ("gpu_layers", c_int), ("mmap", c_bool), ("mlock", c_bool)]


# name -> (restype, argtypes after the LLM* handle).  Source of truth for the ABI on the Python side; the
# CPU test-suite checks include/ctransformers_llm.h and the built .so against this table.
ABI = {
    "ctransformers_llm_delete": (None, []),
    "ctransformers_llm_tokenize": (c_int, [c_char_p, c_bool, c_int_p]),
    "ctransformers_llm_detokenize": (c_char_p, [c_int]),
    "ctransformers_llm_is_eos_token": (c_bool, [c_int]),
    "ctransformers_llm_eos_token_id": (c_int, []),
    "ctransformers_llm_bos_token_id": (c_int, []),
    "ctransformers_llm_vocab_size": (c_int, []),
    "ctransformers_llm_context_length": (c_int, []),
    "ctransformers_llm_architecture": (c_char_p, []),
    "ctransformers_llm_batch_eval": (c_bool, [c_int_p, c_int, c_int, c_int, c_int]),
    "ctransformers_llm_logits_data": (c_float_p, []),
    "ctransformers_llm_logits_size": (c_int, []),
    "ctransformers_llm_embeddings_data": (c_float_p, []),
    "ctransformers_llm_embeddings_size": (c_int, []),
    "ctransformers_llm_sample": (c_int, [c_int_p, c_int, c_int, c_float, c_float, c_float, c_int]),
    "ctransformers_llm_reset": (None, []),
}
ABI_SYMBOLS = ["ctransformers_llm_create"] + list(ABI)


def default_library_path() -> str:
    return str(Path(__file__).parent.resolve() / "lib" / "libctransformers.so")


def find_library(path: Optional[str] = None) -> str:
    """`lib=None` -> this repo's HIP build; anything else is taken as a literal path (reference lib.py:12-15)."""
    path = path or os.environ.get("CTRANSFORMERS_AMD_LIB") or default_library_path()
    if not Path(path).is_file():
        raise OSError(
            "HIP library '%s' not found. Build it with `python __graft_entry__.py build` "
            "(or `make -C ctransformers_amd/csrc`). There is no CPU fallback." % path)
    return path


def load_library(path: Optional[str] = None):
    lib = ctypes.CDLL(find_library(path))
    lib.ctransformers_llm_create.argtypes = [c_char_p, c_char_p, ConfigStruct]
    lib.ctransformers_llm_create.restype = c_void_p
    for name, (restype, argtypes) in ABI.items():
        fn = getattr(lib, name)
        fn.argtypes = [c_void_p] + argtypes
        fn.restype = restype
    return lib


def is_gguf(path) -> bool:
    with open(str(Path(path).resolve()), "rb") as f:
        return f.read(4) == b"GGUF"


class Vector:
    """List-like in-place view of a callee-owned C float array (reference utils.py:13-44): writes go through to
    the library's buffer and persist across property reads until the next eval."""

    def __init__(self, data, size):
        self._data, self._size = data, size

    def _check(self, i):
        if not isinstance(i, int):
            raise TypeError("list index must be integer")
        if not 0 <= i < self._size:
            raise IndexError("list index out of range")

    def __getitem__(self, i):
        self._check(i)
        return self._data[i]

    def __setitem__(self, i, v):
        self._check(i)
        self._data[i] = v

    def __len__(self):
        return self._size

    def __iter__(self):
        for i in range(self._size):
            yield self._data[i]

    def __delitem__(self, i):
        raise NotImplementedError("This operation is not allowed.")

    def insert(self, i, v):
        raise NotImplementedError("This operation is not allowed.")

    def to_numpy(self):
        import numpy as np
        if self._size == 0:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(self._data, shape=(self._size,))


def utf8_split_incomplete(seq: bytes):
    """Split off a trailing run of bytes with the high bit set (reference utils.py:47-56 semantics)."""
    i = len(seq)
    while i > 0 and (seq[i - 1] & 0x80):
        i -= 1
    return seq[:i], seq[i:]


def _get(*values):
    for v in values:
        if v is not None:
            return v


class LLM:
    def __init__(self, model_path: str, model_type: Optional[str] = None, *, config: Optional[Config] = None,
                 lib: Optional[str] = None):
        config = config or Config()
        self._model_path, self._config = model_path, config
        self._llm, self._lib, self._context = None, None, []
        if not Path(model_path).is_file():
            raise ValueError(f"Model path '{model_path}' doesn't exist.")
        if not model_type:
            if not is_gguf(model_path):
                raise ValueError("Unable to detect model type. Please specify a model type using:\n\n"
                                 "  AutoModelForCausalLM.from_pretrained(..., model_type='...')\n\n")
            model_type = "gguf"
        self._lib = load_library(lib)
        self._llm = self._lib.ctransformers_llm_create(model_path.encode(), model_type.encode(), config.to_struct())
        if self._llm is None:
            raise RuntimeError(f"Failed to create LLM '{model_type}' from '{model_path}'.")
        arch = self.ctransformers_llm_architecture().decode()
        self._model_type = arch or model_type

    model_path = property(lambda self: self._model_path)
    model_type = property(lambda self: self._model_type)
    config = property(lambda self: self._config)
    eos_token_id = property(lambda self: self.ctransformers_llm_eos_token_id())
    bos_token_id = property(lambda self: self.ctransformers_llm_bos_token_id())
    pad_token_id = property(lambda self: self.ctransformers_llm_eos_token_id())
    vocab_size = property(lambda self: self.ctransformers_llm_vocab_size())
    context_length = property(lambda self: self.ctransformers_llm_context_length())

    @property
    def logits(self) -> Vector:
        return Vector(self.ctransformers_llm_logits_data(), self.ctransformers_llm_logits_size())

    @property
    def embeddings(self) -> Vector:
        return Vector(self.ctransformers_llm_embeddings_data(), self.ctransformers_llm_embeddings_size())

    def __getattr__(self, name):
        lib, llm = self.__dict__.get("_lib"), self.__dict__.get("_llm")
        if name.startswith("ctransformers_llm_") and lib is not None and hasattr(lib, name):
            return partial(getattr(lib, name), llm)
        raise AttributeError(f"'LLM' object has no attribute '{name}'")

    def tokenize(self, text: str, add_bos_token: Optional[bool] = None) -> List[int]:
        if add_bos_token is None:
            add_bos_token = self.model_type == "llama"
        # the reference allocates len(text)+1 ints (llm.py:335); byte-fallback vocabularies can need one id per UTF-8
        # byte (+BOS, +the leading-space escape), so be generous — the callee does not bounds-check.
        out = (c_int * (len(text.encode()) + 8))()
        n = self.ctransformers_llm_tokenize(text.encode(), add_bos_token, out)
        return out[:n]

    def detokenize(self, tokens: Sequence[int], decode: bool = True) -> Union[str, bytes]:
        if isinstance(tokens, int):
            tokens = [tokens]
        raw = b"".join(self.ctransformers_llm_detokenize(t) for t in tokens)
        if not decode:
            return raw
        text = raw.decode(errors="ignore")
        if list(tokens[:1]) == [self.bos_token_id] and text[:1] == " ":
            text = text[1:]
        return text

    def is_eos_token(self, token: int) -> bool:
        return self.ctransformers_llm_is_eos_token(token)

    def eval(self, tokens: Sequence[int], *, batch_size: Optional[int] = None, threads: Optional[int] = None) -> None:
        """Evaluate `tokens` at position len(context); the hot entry (reference llm.py:379-412)."""
        batch_size = _get(batch_size, self._config.batch_size)
        threads = _get(threads, self._config.threads)
        n_past, n_tokens = len(self._context), len(tokens)
        if n_past + n_tokens > self.context_length:
            logger.warning(f"Number of tokens ({n_past + n_tokens}) exceeded maximum context length "
                           f"({self.context_length}).")
        arr = (c_int * n_tokens)(*tokens)
        if not self.ctransformers_llm_batch_eval(arr, n_tokens, n_past, batch_size, threads):
            raise RuntimeError("Failed to evaluate tokens.")
        self._context.extend(arr)

    def sample(self, *, top_k=None, top_p=None, temperature=None, repetition_penalty=None, last_n_tokens=None,
               seed=None) -> int:
        c = self._config
        top_k, top_p = _get(top_k, c.top_k), _get(top_p, c.top_p)
        temperature = _get(temperature, c.temperature)
        repetition_penalty = _get(repetition_penalty, c.repetition_penalty)
        last_n_tokens, seed = _get(last_n_tokens, c.last_n_tokens), _get(seed, c.seed)
        if last_n_tokens < 0:
            last_n_tokens = self.context_length
        last = self._context[-last_n_tokens:]   # last_n_tokens=0 slices [-0:]: the whole context, as the reference does (llm.py:443)
        arr = (c_int * len(last))(*last)
        return self.ctransformers_llm_sample(arr, len(last), top_k, top_p, temperature, repetition_penalty, seed)

    def reset(self) -> None:
        warnings.warn("`LLM.reset()` method is deprecated since 0.2.27. Please use high-level API.")
        self._context.clear()
        self.ctransformers_llm_reset()

    def __del__(self):
        if self.__dict__.get("_llm") is not None and self.__dict__.get("_lib") is not None:
            self._lib.ctransformers_llm_delete(self._llm)
            self._llm = None

    def prepare_inputs_for_generation(self, tokens: Sequence[int], *, reset: Optional[bool] = None) -> Sequence[int]:
        """Prefix reuse: drop leading tokens already in the context (reference llm.py:470-500)."""
        if not _get(reset, self._config.reset):
            return tokens
        n = min(len(tokens) - 1, len(self._context))
        k = 0
        while k < n and tokens[k] == self._context[k]:
            k += 1
        self._context = self._context[:k]
        return tokens[k:]

    def generate(self, tokens: Sequence[int], *, top_k=None, top_p=None, temperature=None, repetition_penalty=None,
                 last_n_tokens=None, seed=None, batch_size=None, threads=None, reset=None
                 ) -> Generator[int, None, None]:
        tokens = self.prepare_inputs_for_generation(tokens, reset=reset)
        self.eval(tokens, batch_size=batch_size, threads=threads)
        while True:
            token = self.sample(top_k=top_k, top_p=top_p, temperature=temperature,
                                repetition_penalty=repetition_penalty, last_n_tokens=last_n_tokens, seed=seed)
            self.eval([token], batch_size=batch_size, threads=threads)
            if self.is_eos_token(token):
                break
            yield token

    def _stream(self, prompt: str, *, max_new_tokens=None, stop=None, **kw) -> Generator[str, None, None]:
        c = self._config
        max_new_tokens = _get(max_new_tokens, c.max_new_tokens)
        stop = _get(stop, c.stop) or []
        if isinstance(stop, str):
            stop = [stop]
        stop_re = re.compile("|".join(map(re.escape, stop)))
        count, text, pending = 0, "", b""
        for token in self.generate(self.tokenize(prompt), **kw):
            pending += self.detokenize([token], decode=False)
            done, pending = utf8_split_incomplete(pending)
            text += done.decode(errors="ignore")
            if stop:
                m = stop_re.search(text)
                if m:
                    text = text[:m.start()]
                    break
            # hold back the longest suffix that could still grow into a stop sequence
            hold = 0
            for s in stop:
                for i in range(len(s), 0, -1):
                    if text.endswith(s[:i]):
                        hold = max(hold, i)
                        break
            end = len(text) - hold
            if end > 0:
                yield text[:end]
                text = text[end:]
            count += 1
            if count >= max_new_tokens:
                break
        if text:
            yield text

    def __call__(self, prompt: str, *, max_new_tokens=None, top_k=None, top_p=None, temperature=None,
                 repetition_penalty=None, last_n_tokens=None, seed=None, batch_size=None, threads=None, stop=None,
                 stream=None, reset=None) -> Union[str, Generator[str, None, None]]:
        gen = self._stream(prompt, max_new_tokens=max_new_tokens, stop=stop, top_k=top_k, top_p=top_p,
                           temperature=temperature, repetition_penalty=repetition_penalty,
                           last_n_tokens=last_n_tokens, seed=seed, batch_size=batch_size, threads=threads,
                           reset=reset)
        return gen if _get(stream, self._config.stream) else "".join(gen)

    def embed(self, input: Union[str, Sequence[int]], *, batch_size=None, threads=None) -> List[float]:
        if isinstance(input, str):
            input = self.tokenize(input)
        input = self.prepare_inputs_for_generation(input, reset=True)
        self.eval(input, batch_size=batch_size, threads=threads)
        return list(self.embeddings)


_CONFIG_FIELDS = {f.name for f in fields(Config)}


class AutoModelForCausalLM:
    """Local-path subset of reference hub.py:108-200 (no network in scope): a file, or a directory in which the
    smallest *.gguf / *.bin file is picked (reference hub.py:233-252)."""

    @classmethod
    def from_pretrained(cls, model_path_or_repo_id: str, *, model_type: Optional[str] = None,
                        model_file: Optional[str] = None, config: Optional[Config] = None, lib: Optional[str] = None,
                        **kwargs) -> LLM:
        config = config or Config()
        for k, v in kwargs.items():
            if k not in _CONFIG_FIELDS:
                raise TypeError(f"'{k}' is an invalid keyword argument for from_pretrained()")
            setattr(config, k, v)
        p = Path(model_path_or_repo_id)
        if p.is_dir():
            if model_file:
                p = p / model_file
            else:
                cands = [f for f in p.iterdir() if f.is_file() and f.suffix in (".bin", ".gguf")]
                if not cands:
                    raise ValueError(f"No model file found in directory '{model_path_or_repo_id}'")
                p = min(cands, key=lambda f: f.stat().st_size)
        elif not p.is_file():
            raise ValueError(f"Model path '{model_path_or_repo_id}' doesn't exist.")
        return LLM(model_path=str(p), model_type=model_type, config=config, lib=lib)
