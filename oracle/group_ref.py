"""oracle/group_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-Python restatement of the reference's group package, the checker for
gofindthem_amd/csrc/group_host.cpp.  Plain Python loops: rule sets and objects are small.

Follows, statement by statement (paths relative to /root/reference):
  * group/dsl/scanner.go:12-263     tokens, Scan, scanWhitespace, scanOperators, scanTag, scanFieldPath
  * group/dsl/parser.go:24-297      parse, handleDualOp, handleOpenPar, parseTagInfo, GetFields, GetTags
  * group/dsl/expression.go:11-125  ExprType, TagInfo, Expression, Solve
  * group/finder/finder.go:28-196   AddRule(s), GetFieldNames, TagJson/TagObject/TagText, EvaluateRules, Process*
  * group/finder/internal.go:9-119  getRulesInfo (the object walk), isValidateFieldPath

Pinned by the reference's own tables, transcribed to tests/golden/group_*.json (group/dsl/*_test.go,
group/finder/*_test.go).  The finder underneath is injected (`process_text`): tests pass the CPU oracle's
ProcessText, so nothing here touches the GPU.  Nothing under gofindthem_amd/ imports this module.
"""
import json

# group/dsl/scanner.go:12-35
ILLEGAL, EOF, WS, TAG, FIELD_PATH, QUOTATION, OPPAR, CLPAR, AND, OR, NOT = range(11)
TOKEN_NAMES = ["ILLEGAL", "EOF", "WS", "TAG", "FIELD_PATH", "QUOTATION", "OPPAR", "CLPAR", "AND", "OR", "NOT"]

# group/dsl/expression.go:11-17
UNSET_EXPR, AND_EXPR, OR_EXPR, NOT_EXPR, UNIT_EXPR = range(5)
EXPR_NAMES = ["UNSET", "AND", "OR", "NOT", "UNIT"]

_EOF_CH = "\0"   # scanner.go:263


class GroupDslError(Exception):
    pass


def _is_ws(ch):
    return ch in (" ", "\t", "\n")


def _is_letter(ch):
    return ("a" <= ch <= "z") or ("A" <= ch <= "Z")


class Scanner:
    """group/dsl/scanner.go:67-263."""

    def __init__(self, text):
        self.s, self.i = text, 0

    def _read(self):
        if self.i >= len(self.s):
            return _EOF_CH
        ch = self.s[self.i]
        self.i += 1
        return ch

    def _unread(self):
        self.i -= 1

    def scan(self):                                   # scanner.go:78-109
        ch = self._read()
        if _is_ws(ch):
            self._unread()
            return self._scan_whitespace()
        if ch == '"':
            self._unread()
            return self._scan_tag()
        if ch == ":":
            self._unread()
            return self._scan_field_path()
        if _is_letter(ch):
            self._unread()
            return self._scan_operators()
        if ch == "(":
            return OPPAR, "(", None
        if ch == ")":
            return CLPAR, ")", None
        if ch == _EOF_CH:
            return EOF, "", None
        return ILLEGAL, "", "illegal char was found %s" % ch

    def _scan_whitespace(self):                       # scanner.go:112-131
        buf = [self._read()]
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                break
            if not _is_ws(ch):
                self._unread()
                break
            buf.append(ch)
        return WS, "".join(buf), None

    def _scan_operators(self):                        # scanner.go:134-172
        ch = self._read()
        if not _is_letter(ch):
            return ILLEGAL, "", "fail to scan operator: expected letter but found %s" % ch
        buf = [ch]
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                break
            if not _is_letter(ch):
                self._unread()
                break
            buf.append(ch)
        lit = "".join(buf)
        up = lit.upper()
        if up == "AND":
            return AND, lit, None
        if up == "OR":
            return OR, lit, None
        if up == "NOT":
            return NOT, lit, None
        return ILLEGAL, "", "failed to scan operator: unexpected operator '%s' found" % lit

    def _scan_tag(self):                              # scanner.go:177-210
        ch = self._read()
        if ch != '"':
            return ILLEGAL, "", 'fail to scan tag: expected " but found %s' % ch
        buf = []
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                return ILLEGAL, "", "fail to scan tag: expected ':' but found EOF"
            if ch == "\\":
                esc = self._read()
                if esc in ("\\", '"', ":"):
                    buf.append(esc)
                else:
                    return ILLEGAL, "", "fail to scan tag: invalid escaped char %s" % esc
                continue
            if ch == ":":
                self._unread()
                break
            if ch == '"':
                break
            buf.append(ch)
        return TAG, "".join(buf).strip(" "), None

    def _scan_field_path(self):                       # scanner.go:215-244
        ch = self._read()
        if ch != ":":
            return ILLEGAL, "", "fail to scan field: expected ':' but found %s" % ch
        buf = []
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                return ILLEGAL, "", "fail to scan field: expected '\"' but found EOF"
            if ch == "\\":
                esc = self._read()
                if esc in ("\\", '"'):
                    buf.append(esc)
                else:
                    return ILLEGAL, "", "fail to scan field: invalid escaped char %s" % esc
                continue
            if ch == '"':
                break
            buf.append(ch)
        return FIELD_PATH, "".join(buf).strip(" "), None


def tokens(text):
    """token stream up to and including EOF or the first error (what scanner_test.go walks)."""
    sc, out = Scanner(text), []
    while True:
        tok, lit, err = sc.scan()
        out.append({"Tok": TOKEN_NAMES[tok], "Lit": lit, "Err": err})
        if err is not None or tok == EOF:
            return out


class Expression:
    """group/dsl/expression.go:46-51."""
    __slots__ = ("LExpr", "RExpr", "Type", "Name", "FieldPath")

    def __init__(self, typ=UNSET_EXPR, name="", field_path=""):
        self.LExpr = self.RExpr = None
        self.Type, self.Name, self.FieldPath = typ, name, field_path

    def to_obj(self):
        d = {"Type": EXPR_NAMES[self.Type]}
        if self.Type == UNIT_EXPR:
            d["Tag"] = {"Name": self.Name, "FieldPath": self.FieldPath}
        if self.LExpr is not None:
            d["LExpr"] = self.LExpr.to_obj()
        if self.RExpr is not None:
            d["RExpr"] = self.RExpr.to_obj()
        return d


class Parser:
    """group/dsl/parser.go:10-297."""

    def __init__(self, text):
        self.s = Scanner(text)
        self.buf_tok, self.buf_lit, self.unscanned = ILLEGAL, "", False
        self.par_count = 0
        self.tags, self.fields = [], []

    def _scan(self):                                  # parser.go:204-219
        if self.unscanned:
            self.unscanned = False
            return self.buf_tok, self.buf_lit
        tok, lit, err = self.s.scan()
        if err is not None:
            raise GroupDslError(err)
        self.buf_tok, self.buf_lit = tok, lit
        return tok, lit

    def _unscan(self):
        self.unscanned = True

    def _scan_ignore_ws(self):                        # parser.go:226-235
        tok, lit = self._scan()
        if tok == WS:
            tok, lit = self._scan()
        return tok, lit

    def _note(self, name, field):
        if name not in self.tags:
            self.tags.append(name)
        if field != "" and field not in self.fields:
            self.fields.append(field)

    def _parse_tag_info(self):                        # parser.go:252-278
        tok, lit = self._scan_ignore_ws()
        if tok != TAG:
            raise GroupDslError("invalid expression: Expecting TAG but found %s" % TOKEN_NAMES[tok])
        if lit == "":
            raise GroupDslError("invalid expression: Found empty TAG")
        ntok, nlit = self._scan_ignore_ws()
        if ntok != FIELD_PATH:
            self._unscan()
            return lit, ""
        return lit, nlit

    def _handle_open_par(self):                       # parser.go:238-249
        lvl = self.par_count
        self.par_count += 1
        e = self._parse()
        if self.par_count != lvl:
            raise GroupDslError("invalid expression: Unexpected '('")
        return e

    def _handle_dual_op(self, exp, typ):              # parser.go:178-201
        if exp.LExpr is None:
            raise GroupDslError("invalid expression: no left expression was found for %s" % EXPR_NAMES[typ])
        if exp.RExpr is None:
            exp.Type = typ
            return exp
        up = Expression(typ)
        up.LExpr = exp
        tok, _ = self._scan_ignore_ws()
        if tok == OPPAR:
            up.RExpr = self._handle_open_par()
        else:
            self._unscan()
        return up

    @staticmethod
    def _attach(exp, child):
        if exp.LExpr is None:
            exp.LExpr = child
        else:
            exp.RExpr = child

    def parse(self):
        return self._parse()

    def _parse(self):                                 # parser.go:41-175
        exp = Expression()
        while True:
            tok, lit = self._scan_ignore_ws()
            if tok == OPPAR:
                self._attach(exp, self._handle_open_par())
            elif tok == TAG:
                self._unscan()
                name, field = self._parse_tag_info()
                self._attach(exp, Expression(UNIT_EXPR, name, field))
                self._note(name, field)
            elif tok == AND:
                exp = self._handle_dual_op(exp, AND_EXPR)
            elif tok == OR:
                exp = self._handle_dual_op(exp, OR_EXPR)
            elif tok == NOT:
                ntok, _ = self._scan_ignore_ws()
                neg = Expression(NOT_EXPR)
                if ntok == TAG:
                    self._unscan()
                    name, field = self._parse_tag_info()
                    neg.RExpr = Expression(UNIT_EXPR, name, field)
                    self._note(name, field)
                elif ntok == OPPAR:
                    neg.RExpr = self._handle_open_par()
                else:
                    raise GroupDslError("invalid expression: Unexpected token '%s' after NOT" % TOKEN_NAMES[ntok])
                self._attach(exp, neg)
            elif tok in (CLPAR, EOF):
                if tok == CLPAR:
                    self.par_count -= 1
                if self.par_count < 0:
                    raise GroupDslError("invalid expression: unexpected EOF found. Extra closing parentheses: %d"
                                        % -self.par_count)
                fin = exp
                if exp.Type == UNSET_EXPR:
                    if exp.RExpr is not None:
                        fin = exp.RExpr
                    elif exp.LExpr is not None:
                        fin = exp.LExpr
                    else:
                        raise GroupDslError("invalid expression: unexpected EOF found")
                if fin.Type in (AND_EXPR, OR_EXPR) and fin.RExpr is None:
                    raise GroupDslError("invalid expression: incomplete expression %s" % EXPR_NAMES[fin.Type])
                return fin
            else:
                raise GroupDslError("invalid expression: Unexpected operator was found (%d = '%s')" % (tok, lit))


def parse(text):
    """-> (Expression, tags, fields); raises GroupDslError with the reference's text."""
    p = Parser(text)
    e = p.parse()
    return e, p.tags, p.fields


def solve(exp, tagmap):
    """Expression.Solve (expression.go:61-125).  tagmap: {tag: {field: set/None} or None}."""
    if exp.Type == UNIT_EXPR:
        if exp.Name in tagmap:
            if exp.FieldPath == "":
                return True
            for fp in (tagmap[exp.Name] or {}):
                if fp.startswith(exp.FieldPath):
                    return True
        return False
    if exp.Type in (AND_EXPR, OR_EXPR):
        if exp.LExpr is None or exp.RExpr is None:
            raise GroupDslError("%s statement do not have right or left expression" % EXPR_NAMES[exp.Type])
        l, r = solve(exp.LExpr, tagmap), solve(exp.RExpr, tagmap)
        return (l and r) if exp.Type == AND_EXPR else (l or r)
    if exp.Type == NOT_EXPR:
        if exp.RExpr is None:
            raise GroupDslError("NOT statement do not have expression")
        return not solve(exp.RExpr, tagmap)
    raise GroupDslError("unable to process expression type %d" % exp.Type)


def is_valid_field_path(field_path, include_paths, exclude_paths):
    """isValidateFieldPath (internal.go:99-119)."""
    for x in exclude_paths or []:
        if field_path.startswith(x):
            return False
    if include_paths:
        return any(field_path.startswith(x) for x in include_paths)
    return True


def exported_fields(obj):
    """the fields a Go struct walk would visit: exported names only (internal.go:47-49 CanInterface)."""
    return [(k, v) for k, v in vars(obj).items() if k[:1].isupper()]


class GroupFinder:
    """group/finder/finder.go:12-196 over an injected `process_text(text) -> [(tag, expression string)]`."""

    def __init__(self, process_text):
        self.process_text = process_text
        self.rules = {}            # name -> [(expression string, Expression)]
        self.fields, self.tags = set(), set()

    def add_rule(self, name, expressions):            # finder.go:45-66
        for raw in expressions:
            e, tags, fields = parse(raw)
            self.rules.setdefault(name, []).append((raw, e))
            self.tags.update(tags)
            self.fields.update(fields)

    def add_rules(self, rules_by_name):
        for k, v in rules_by_name.items():
            self.add_rule(k, v)

    def get_field_names(self):
        return sorted(self.fields)

    def _walk(self, data, field_name, inc, exc, out):  # internal.go:9-97
        if isinstance(data, str):
            if not is_valid_field_path(field_name, inc, exc):
                return
            for tag, expr in self.process_text(data):
                out.setdefault(tag, {}).setdefault(field_name, set()).add(expr)
        elif isinstance(data, dict):
            if any(not isinstance(k, str) for k in data):
                return                                 # a Go map whose key type is not string is not walked
            for k, v in data.items():
                self._walk(v, k if field_name == "" else field_name + "." + k, inc, exc, out)
        elif isinstance(data, (list, tuple)):
            for i, v in enumerate(data):
                fn = "index(%d)" % i
                self._walk(v, fn if field_name == "" else field_name + "." + fn, inc, exc, out)
        elif hasattr(data, "__dict__") and not isinstance(data, (int, float, bool)):
            for k, v in exported_fields(data):
                self._walk(v, k if field_name == "" else field_name + "." + k, inc, exc, out)

    def tag_object(self, data, include_paths=None, exclude_paths=None):
        out = {}
        self._walk(data, "", include_paths, exclude_paths, out)
        return out

    def tag_json(self, raw, include_paths=None, exclude_paths=None):
        return self.tag_object(json.loads(raw), include_paths, exclude_paths)

    def tag_text(self, text):                          # finder.go:106-121
        return {tag: sorted(fields[""]) for tag, fields in self.tag_object(text).items() if fields.get("")}

    def evaluate_rules(self, tagmap):                  # finder.go:118-137
        out = {}
        for name, wrappers in self.rules.items():
            for raw, e in wrappers:
                if solve(e, tagmap):
                    out.setdefault(name, []).append(raw)
        return out

    def process_json(self, raw, include_paths=None, exclude_paths=None):
        return self.evaluate_rules(self.tag_json(raw, include_paths, exclude_paths))

    def process_object(self, obj, include_paths=None, exclude_paths=None):
        return self.evaluate_rules(self.tag_object(obj, include_paths, exclude_paths))

    def process_text_rules(self, text):                # GroupFinder.ProcessText (finder.go:186-196)
        return self.evaluate_rules(self.tag_object(text))
