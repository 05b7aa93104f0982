"""MI355X-native quantized inference core behind the ctransformers C ABI."""
from .llm import LLM, Config, AutoModelForCausalLM, Vector  # noqa: F401

__version__ = "0.1.0"
