"""Python mirror of the reference's group finder (group/finder/finder.go) over libgft.so's gft_group_* C ABI
(SURVEY.md 8(f) row 2).  Same names and error behaviour; Go `error` values surface as GroupFinderError.

The object walk of group/finder/internal.go is the C++ side's job (csrc/group_host.cpp): every string leaf of every
document of a call goes through the finder as ONE batch on the GPU.  Python objects are handed over as JSON, which
is exactly the shape the walk understands (dict -> "key", list -> "index(i)", str -> a leaf); for plain objects only
attributes with an upper-case first letter are visible, like exported Go struct fields (internal.go:47-49).
"""
import ctypes as C
import json

from . import _lib
from .engine import pack
from .finder import Finder


class GroupFinderError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _jsonable(obj):
    """the part of a Python value the reference's reflect walk would see, as JSON-compatible data"""
    if isinstance(obj, str):
        return obj
    if isinstance(obj, dict):
        if any(not isinstance(k, str) for k in obj):
            return None                       # a Go map whose key type is not string is not walked (internal.go:62-64)
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, (bool, int, float)) or obj is None:
        return None                           # not taggable; the value itself is irrelevant
    if hasattr(obj, "__dict__"):
        return {k: _jsonable(v) for k, v in vars(obj).items() if k[:1].isupper()}
    return None


def _json_call(fn, *args):
    """the out/cap/needed convention of the JSON-returning entry points"""
    need = C.c_uint64(0)
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        rc = fn(*args, C.cast(buf, C.c_void_p), cap, C.byref(need))
        if rc == _lib.GFT_E_INVALID and need.value > cap:
            cap = int(need.value)
            continue
        return rc, buf.value.decode("utf-8", "replace")


def dsl_parse(expr):
    """group/dsl Parser.Parse: {"tree":..,"tags":[..],"fields":[..]} or {"error": <reference text>} (host only)"""
    e = expr.encode("utf-8")
    rc, doc = _json_call(_lib.load().gft_group_dsl_parse, e, len(e))
    assert rc == 0
    return json.loads(doc)


def dsl_tokens(expr):
    e = expr.encode("utf-8")
    rc, doc = _json_call(_lib.load().gft_group_dsl_tokens, e, len(e))
    assert rc == 0
    return json.loads(doc)


class GroupFinder:
    """group/finder.GroupFinder.  NewFinder(findthem) / NewFinderWithRules(findthem, rulesByName)."""

    def __init__(self, findthem: Finder):
        self._L = _lib.load()
        self.findthem = findthem
        h = C.c_void_p()
        rc = self._L.gft_group_create(C.byref(h), findthem._h)
        if rc != 0:
            raise GroupFinderError(rc, "gft_group_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.gft_group_destroy(self._h)
            self._h = None

    __del__ = close

    def _err(self, rc):
        return GroupFinderError(rc, self._L.gft_group_last_error(self._h).decode("utf-8", "replace"))

    # -- rules (finder.go:45-85) -----------------------------------------------------------------------
    def AddRule(self, ruleName, expressions):
        n = ruleName.encode("utf-8")
        for raw in expressions:
            e = raw.encode("utf-8")
            rc = self._L.gft_group_add_rule(self._h, n, len(n), e, len(e))
            if rc != 0:
                raise self._err(rc)

    def AddRules(self, rulesByName):
        for k, v in rulesByName.items():
            self.AddRule(k, v)

    def state(self):
        """{"rules": {name: [{"ExpressionString", "Expression"}]}, "fields": [..], "tags": [..]}"""
        rc, doc = _json_call(self._L.gft_group_state, self._h)
        if rc != 0:
            raise self._err(rc)
        return json.loads(doc)

    def GetFieldNames(self):
        return self.state()["fields"]

    # -- tagging + rules over batches -------------------------------------------------------------------
    def _process(self, raws, includePaths, excludePaths, what):
        raws = [r.encode("utf-8") if isinstance(r, str) else bytes(r) for r in raws]
        blob, off = pack(raws)
        inc = json.dumps(list(includePaths)).encode() if includePaths else None
        exc = json.dumps(list(excludePaths)).encode() if excludePaths else None
        need = C.c_uint64(0)
        cap = max(1 << 16, 2 * int(blob.size))
        buf = C.create_string_buffer(cap)
        rc = self._L.gft_group_process_jsons(self._h, blob.ctypes.data, off.ctypes.data, len(raws), inc, len(inc) if inc else 0,
                                             exc, len(exc) if exc else 0, what, C.cast(buf, C.c_void_p), cap, C.byref(need))
        if rc == _lib.GFT_E_INVALID and need.value > cap:      # the library kept the document: fetch it, no second run
            cap = int(need.value)
            buf = C.create_string_buffer(cap)
            rc = self._L.gft_group_last_result(self._h, C.cast(buf, C.c_void_p), cap, C.byref(need))
        if rc != 0:
            raise self._err(rc)
        return json.loads(buf.value.decode("utf-8", "replace"))

    def ProcessJsons(self, rawJsons, includePaths=None, excludePaths=None):
        """batch extension of ProcessJson: one {"rules": {rule: [expressions]}} or {"error": ..} per document"""
        return self._process(rawJsons, includePaths, excludePaths, 0)

    def TagJsons(self, rawJsons, includePaths=None, excludePaths=None):
        return self._process(rawJsons, includePaths, excludePaths, 1)

    @staticmethod
    def _one(res, key):
        if "error" in res:
            raise GroupFinderError(_lib.GFT_E_ENGINE, res["error"])
        return res[key]

    def TagJson(self, data, includePaths=None, excludePaths=None):                 # finder.go:80-92
        return self._one(self.TagJsons([data], includePaths, excludePaths)[0], "tags")

    def TagObject(self, data, includePaths=None, excludePaths=None):               # finder.go:95-103
        return self.TagJson(json.dumps(_jsonable(data)), includePaths, excludePaths)

    def TagText(self, data):                                                       # finder.go:106-121
        return {tag: fields[""] for tag, fields in self.TagObject(data).items() if fields.get("")}

    def EvaluateRules(self, matchedExpByFieldByTag):                               # finder.go:118-137
        doc = json.dumps({t: ({f: sorted(v or ()) for f, v in fs.items()} if fs else None)
                          for t, fs in matchedExpByFieldByTag.items()}).encode()
        rc, out = _json_call(self._L.gft_group_evaluate, self._h, doc, len(doc))
        if rc != 0:
            raise self._err(rc)
        return json.loads(out)

    def ProcessJson(self, rawJson, includePaths=None, excludePaths=None):          # finder.go:160-172
        return self._one(self.ProcessJsons([rawJson], includePaths, excludePaths)[0], "rules")

    def ProcessObject(self, obj, includePaths=None, excludePaths=None):            # finder.go:180-190
        return self.ProcessJson(json.dumps(_jsonable(obj)), includePaths, excludePaths)

    def ProcessText(self, data):                                                   # finder.go:186-196
        return self.ProcessObject(data)

    def last_batch(self):
        """(string leaves, text bytes) the last call sent through the finder"""
        a, b = C.c_uint64(), C.c_uint64()
        self._L.gft_group_last_batch(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)


def NewFinder(findthem):
    return GroupFinder(findthem)


def NewFinderWithRules(findthem, rulesByName):
    g = GroupFinder(findthem)
    g.AddRules(rulesByName)
    return g


__all__ = ["GroupFinder", "GroupFinderError", "NewFinder", "NewFinderWithRules", "dsl_parse", "dsl_tokens"]
