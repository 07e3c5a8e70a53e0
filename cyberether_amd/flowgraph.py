"""Flowgraph YAML loader: builds the `graph:` section of a CyberEther flowgraph file
on the HIP device, unmodified.

On-disk format (examples/flowgraphs/spectrum-analyzer.yml:10-71): a list of nodes
``{name, module: <block type>, device, runtime, provider, config{...}, input{port:
'${graph.<node>.output.<port>}'}, meta{...}}``.  A node names a BLOCK; blocks expand into modules
exactly as the reference's block_impl.cc files do (spectrum_engine/block_impl.cc:120-217,
filter/block_impl.cc:350-582, decimator/block_impl.cc:140-207, slice/block_impl.cc:55-70); every
other block used by the examples wraps one module of the same name.

What this loader does NOT build is everything SURVEY §8 puts out of scope: `meta` (node editor
geometry) is ignored, `note` blocks are dropped, host sinks (`audio`) and decoders outside the path
(`adsb`) are recorded in ``Flowgraph.skipped`` with the tensor they would have consumed, and the
`soapy` source becomes the HBM-resident ring source with the same output contract
(soapy/module_impl.cc:197-201: CF32[numberOfBatches, numberOfTimeSamples], batchAxis 0, sampleAxis 1,
attributes frequency / sampleRate) that the caller feeds with `Flowgraph.feed()`.

``device:`` is overridden to ``hip`` for every node (this library has no other device);
``provider:`` is honoured where a module registers more than one ("fast" amplitude/range, and the
"fast" Filter block = one direct-form FIR kernel).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import jetstream as js

_REF = re.compile(r"^\$\{graph\.([^.}]+)\.output\.([^.}]+)\}$")

# UI-only keys of the visualisation blocks (lineplot/block.hh, waterfall/block.hh): no compute meaning
_UI_KEYS = {"thickness", "numberOfHorizontalLines", "numberOfVerticalLines", "interpolate",
            "zoom", "offset", "translation", "viewSize"}

# block type -> (module type, {block output port: module output port})
_SIMPLE = {
    "window": ("window", {"window": "window"}),
    "invert": ("invert", {"signal": "signal"}),
    "multiply": ("multiply", {"product": "product"}),
    "multiply_constant": ("multiply_constant", {"product": "product"}),
    "add": ("add", {"sum": "sum"}),
    "fft": ("fft", {"signal": "signal"}),
    "amplitude": ("amplitude", {"signal": "signal"}),
    "range": ("range", {"signal": "signal"}),
    "agc": ("agc", {"signal": "signal"}),
    "cast": ("cast", {"buffer": "buffer"}),
    "reshape": ("reshape", {"buffer": "buffer"}),
    "expand_dims": ("expand_dims", {"buffer": "buffer"}),
    "squeeze_dims": ("squeeze_dims", {"buffer": "buffer"}),
    "duplicate": ("duplicate", {"buffer": "buffer"}),
    "pad": ("pad", {"padded": "padded"}),
    "unpad": ("unpad", {"unpadded": "unpadded", "pad": "pad"}),
    "fold": ("fold", {"buffer": "buffer"}),
    "overlap_add": ("overlap_add", {"buffer": "buffer"}),
    "phase_correction": ("phase_correction", {"signal": "signal"}),
    "arithmetic": ("arithmetic", {"buffer": "buffer"}),
    "fm": ("fm", {"signal": "signal"}),
    "am": ("am", {"signal": "signal"}),
    "signal_axes": ("signal_axes", {"buffer": "buffer"}),
    "ones_tensor": ("ones_tensor", {"buffer": "buffer"}),
    "signal_generator": ("signal_generator", {"signal": "signal"}),
    "lineplot": ("lineplot", {}),
    "waterfall": ("waterfall", {}),
    "spectrogram": ("spectrogram", {}),
}
_DROPPED = {"note": "documentation node"}
_SKIPPED = {"audio": "host audio sink (resampler + sound device): outside the device path, SURVEY §8",
            "adsb": "ADS-B decoder: not on the north-star path",
            "file_writer": "host file sink", "websocket": "network sink", "constellation": "render surface",
            "psk_demod": "not on the north-star path", "rrc_filter": "not on the north-star path"}


def _scalar(v):
    """YAML scalars arrive typed, except numbers the reference serialised as strings ('1e+08')."""
    if isinstance(v, str):
        s = v.strip()
        try:
            if re.fullmatch(r"[+-]?\d+", s):
                return int(s)
            return float(s)
        except ValueError:
            return v
    return v


def _number_list(v) -> List[float]:
    if isinstance(v, (list, tuple)):
        return [float(x) for x in v]
    if isinstance(v, (int, float)):
        return [float(v)]
    body = str(v).strip()
    if body.startswith("["):
        body = body[1:-1]
    return [float(t) for t in body.replace(",", " ").split()]


class FlowgraphError(RuntimeError):
    pass


class Node:
    def __init__(self, name: str, block: str):
        self.name, self.block = name, block
        self.modules: List[js.Module] = []
        self.outputs: Dict[str, js.Tensor] = {}
        self.impl = None  # SpectrumEngine / Filter / Decimator object for the composite blocks


class Flowgraph:
    """``Flowgraph(path_or_text)`` parses and instantiates; ``runtime()`` returns a ``js.Runtime`` over all
    modules in creation order.  ``ring_slots``: depth of the device ring behind every `soapy` node."""

    def __init__(self, source: str, ring_slots: int = 4, provider: Optional[str] = None,
                 instantiate: bool = True):
        """instantiate=False only parses and orders the graph (``self.plan``): no device needed."""
        import os
        import yaml
        text = open(source).read() if os.path.exists(source) else source
        doc = yaml.safe_load(text)
        if not isinstance(doc, dict) or "graph" not in doc:
            raise FlowgraphError("not a flowgraph file: no 'graph:' section")
        self.title = doc.get("title", "")
        self.version = doc.get("version")
        self.ring_slots = ring_slots
        self.provider_override = provider
        self.nodes: Dict[str, Node] = {}
        self.skipped: Dict[str, dict] = {}
        self.dropped: List[str] = []
        self.order: List[str] = []
        self.plan: List[dict] = []  # topologically ordered: {name, block, inputs, status}
        entries = doc["graph"] or []
        if isinstance(entries, dict):  # older files key the nodes by name
            entries = [dict(v, name=k) for k, v in entries.items()]
        pending = []
        for e in entries:
            if "name" not in e or "module" not in e:
                raise FlowgraphError(f"graph node without name/module: {e}")
            if e["module"] in _DROPPED:
                self.dropped.append(e["name"])
                continue
            pending.append(e)
        # instantiate in dependency order (the file order is the editor's, not topological)
        done = set()
        composite = {"soapy", "spectrum_engine", "filter", "filter_engine", "decimator", "slice", "filter_taps", "squelch", "flatten",
                     "permutation"}
        while pending:
            progressed = False
            for e in list(pending):
                refs = self._refs(e)
                if all(n in done for n, _ in refs.values()):
                    block = e["module"]
                    status = ("skipped" if block in _SKIPPED else
                              "ok" if block in _SIMPLE or block in composite else "unsupported")
                    self.plan.append({"name": e["name"], "block": block, "status": status,
                                      "inputs": {p: f"{n}.{o}" for p, (n, o) in refs.items()}})
                    if instantiate:
                        self._instantiate(e, refs)
                    done.add(e["name"])
                    pending.remove(e)
                    progressed = True
            if not progressed:
                names = [e["name"] for e in pending]
                raise FlowgraphError(f"unresolved or cyclic inputs among nodes {names}")

    # -- parsing helpers ------------------------------------------------------------------------
    @staticmethod
    def _refs(entry) -> Dict[str, Tuple[str, str]]:
        out = {}
        for port, ref in (entry.get("input") or {}).items():
            m = _REF.match(str(ref).strip())
            if not m:
                raise FlowgraphError(f"node '{entry['name']}': cannot parse input reference {ref!r}")
            out[port] = (m.group(1), m.group(2))
        return out

    def _resolve(self, entry, refs) -> Dict[str, js.Tensor]:
        tensors = {}
        for port, (node, out_port) in refs.items():
            if node in self.skipped:
                raise FlowgraphError(f"node '{entry['name']}' reads from skipped node '{node}' "
                                     f"({self.skipped[node]['reason']})")
            outs = self.nodes[node].outputs
            if out_port not in outs:
                raise FlowgraphError(f"node '{node}' has no output '{out_port}' (has {sorted(outs)})")
            tensors[port] = outs[out_port]
        return tensors

    # -- block expansion ------------------------------------------------------------------------
    def _instantiate(self, entry, refs):
        name, block = entry["name"], entry["module"]
        cfg = {k: _scalar(v) for k, v in (entry.get("config") or {}).items() if k not in _UI_KEYS}
        provider = self.provider_override or entry.get("provider", "generic")
        if block in _SKIPPED:
            consumed = {p: f"{n}.{o}" for p, (n, o) in refs.items()}
            self.skipped[name] = {"block": block, "reason": _SKIPPED[block], "inputs": consumed}
            return
        inputs = self._resolve(entry, refs)
        node = Node(name, block)
        if block == "soapy":
            batches = int(cfg.get("numberOfBatches", 8))
            samples = int(cfg.get("numberOfTimeSamples", 8192))
            m = js.Module("ring_source", {"batches": batches, "samples": samples,
                                          "slots": self.ring_slots}, {}, name)
            out = m.output("buffer")
            out.set_attribute("sampleRate", float(np.float32(cfg.get("sampleRate", 2.0e6))))
            out.set_attribute("frequency", float(np.float32(cfg.get("frequency", 96.9e6))))
            node.modules, node.outputs = [m], {"signal": out}
        elif block == "spectrum_engine":
            eng = js.SpectrumEngine(inputs["buffer"], enable_scale=bool(cfg.get("enableScale", False)),
                                    range_min=float(cfg.get("rangeMin", -120.0)),
                                    range_max=float(cfg.get("rangeMax", 0.0)), name=name,
                                    provider=provider if provider in ("generic", "fast") else "generic",
                                    enable_agc=bool(cfg.get("enableAgc", False)))
            node.impl, node.modules, node.outputs = eng, eng.modules, {"buffer": eng.buffer}
        elif block == "filter":
            heads = int(cfg.get("heads", 1))
            center = (_number_list(cfg.get("center", [0.0])) + [0.0] * heads)[:heads]
            flt = js.Filter(inputs["signal"], float(cfg.get("sampleRate", 2.0e6)),
                            float(cfg.get("bandwidth", 1.0e6)), center, int(cfg.get("taps", 101)),
                            heads, name=name,
                            provider=provider if provider in ("generic", "fast") else "generic")
            node.impl, node.modules, node.outputs = flt, flt.modules, {"buffer": flt.buffer}
        elif block == "filter_engine":  # dsp/filter_engine/block_impl.cc: external coefficient tensor
            eng = js.FilterEngine(inputs["signal"], inputs["filter"], name=name)
            node.impl, node.modules, node.outputs = eng, eng.modules, {"buffer": eng.buffer}
        elif block == "squelch":  # dsp/squelch/block_impl.cc: one module, ports signal -> signal
            sq = js.Module("squelch", {"threshold": float(cfg.get("threshold", 0.1))},
                           {"signal": inputs["signal"]}, name + ".squelch")
            node.modules, node.outputs = [sq], {"signal": sq.output("signal")}
        elif block == "decimator":
            dec = js.Decimator(inputs["buffer"], int(cfg.get("ratio", 4)), name=name)
            node.impl, node.modules, node.outputs = dec, dec.modules, {"buffer": dec.buffer}
        elif block == "slice":
            contiguous = bool(cfg.get("contiguous", False))
            sl = js.Module("slice", {"slice": str(cfg.get("slice", "[...]"))}, inputs, name + ".slice")
            node.modules, out = [sl], sl.output("buffer")
            if contiguous:  # slice/block_impl.cc:60-64
                dup = js.Module("duplicate", {}, {"buffer": out}, name + ".duplicate")
                node.modules.append(dup)
                out = dup.output("buffer")
            node.outputs = {"buffer": out}
        elif block == "filter_taps":
            heads = int(cfg.get("heads", 1))
            center = (_number_list(cfg.get("center", [0.0])) + [0.0] * heads)[:heads]
            m = js.Module("filter_taps", {"sampleRate": float(np.float32(cfg.get("sampleRate", 2.0e6))),
                                          "bandwidth": float(np.float32(cfg.get("bandwidth", 1.0e6))),
                                          "center": [float(np.float32(c)) for c in center],
                                          "taps": int(cfg.get("taps", 101))}, {}, name)
            node.modules, node.outputs = [m], {"coeffs": m.output("coeffs")}
        elif block in ("flatten", "permutation"):
            # flatten/block_impl.cc:50-60 copies BEFORE the view (flatten needs dense input),
            # permutation/block_impl.cc:56-64 copies AFTER it (densifies the strided view)
            contiguous = bool(cfg.pop("contiguous", False))
            src, mods = inputs["buffer"], []
            if contiguous and block == "flatten":
                mods.append(js.Module("duplicate", {}, {"buffer": src}, name + ".duplicate"))
                src = mods[-1].output("buffer")
            mods.append(js.Module(block, cfg, {"buffer": src}, name + "." + block))
            src = mods[-1].output("buffer")
            if contiguous and block == "permutation":
                mods.append(js.Module("duplicate", {}, {"buffer": src}, name + ".duplicate"))
                src = mods[-1].output("buffer")
            node.modules, node.outputs = mods, {"buffer": src}
        elif block in _SIMPLE:
            mtype, ports = _SIMPLE[block]
            # `complexOutput` passes through to the module (fft/block_impl.cc:42-50): it selects the N/2+1
            # complex bins for a forward transform of REAL input and is ignored for CF32 input, exactly as
            # the reference's module does (fft/module_impl.cc:33-38).
            kwargs = {"provider": provider} if mtype in ("amplitude", "range") and provider == "fast" else {}
            m = js.Module(mtype, cfg, inputs, name, **kwargs)
            node.modules = [m]
            node.outputs = {bp: m.output(mp) for bp, mp in ports.items()}
        else:
            raise FlowgraphError(f"node '{name}': block type '{block}' is not implemented on the HIP "
                                 f"device (implemented: {sorted(set(_SIMPLE) | {'soapy', 'spectrum_engine', 'filter', 'filter_engine', 'decimator', 'slice', 'filter_taps', 'flatten', 'permutation', 'squelch'})})")
        self.nodes[name] = node
        self.order.append(name)

    # -- use ------------------------------------------------------------------------------------
    @property
    def modules(self) -> List[js.Module]:
        return [m for n in self.order for m in self.nodes[n].modules]

    def runtime(self, graph: bool = True, fuse: bool = True, **flags) -> js.Runtime:
        return js.Runtime(self.modules, graph=graph, fuse=fuse, **flags)

    def output(self, node: str, port: str) -> js.Tensor:
        return self.nodes[node].outputs[port]

    def module(self, node: str, index: int = -1) -> js.Module:
        return self.nodes[node].modules[index]

    def sources(self) -> List[str]:
        return [n for n in self.order if self.nodes[n].block == "soapy"]

    def feed(self, node: str, samples: np.ndarray, slot: Optional[int] = None):
        """Uploads CF32[batches, samples] into one ring slot of a `soapy` stand-in (all slots when
        slot is None), i.e. what the SDR thread's ring-buffer pop would have delivered."""
        buf = self.nodes[node].outputs["signal"]
        data = np.ascontiguousarray(samples, np.complex64)
        for s in (range(self.ring_slots) if slot is None else [slot]):
            buf.ring_select(s).copy_from(data)
        buf.ring_select(0)  # the slot a fresh runtime exposes first
