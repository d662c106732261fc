"""oracle/dsl_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-Python restatement of the reference's DSL front-end, used only to hand the CPU oracle
(oracle/ac_oracle.cpp) the same expression trees the Go parser would build.  Build-time work
(once per expression), so plain Python loops are fine.

Follows, statement by statement (paths relative to /root/reference):
  * dsl/scanner.go:14-34,79-250   tokens, Scan, scanWhitespace, scanOperators, scanKeyword
  * dsl/parser.go:28-315          NewParser, parse, handleDualOp, handleOpenPar, addLiteralToSet
  * dsl/expression.go:9-48        ExprType / Expression

Pinned by the reference's own tables, transcribed to tests/golden/{scanner,parser,solver}.json
(dsl/scanner_test.go:19-104, dsl/parser_test.go:22-431, dsl/expression_test.go:21-313).
Nothing under gofindthem_amd/ imports this module.
"""

# dsl/scanner.go:14-34
ILLEGAL, EOF, WS, KEYWORD, QUOTATION, OPPAR, CLPAR, AND, OR, NOT, INORD, REGEX = range(12)
TOKEN_NAMES = ["ILLEGAL", "EOF", "WS", "KEYWORD", "QUOTATION", "OPPAR", "CLPAR",
               "AND", "OR", "NOT", "INORD", "REGEX"]

# dsl/expression.go:9-18
UNSET_EXPR, AND_EXPR, OR_EXPR, NOT_EXPR, UNIT_EXPR, INORD_EXPR = range(6)
EXPR_NAMES = ["UNSET", "AND", "OR", "NOT", "UNIT", "INORD"]

_EOF_CH = "\0"  # dsl/scanner.go:250  (rune(0))


class DslError(Exception):
    pass


def _is_ws(ch):
    return ch in (" ", "\t", "\n")


def _is_letter(ch):
    return ("a" <= ch <= "z") or ("A" <= ch <= "Z")


class Scanner:
    """dsl/scanner.go:69-250."""

    def __init__(self, text):
        self.s = text
        self.i = 0

    def _read(self):
        if self.i >= len(self.s):
            return _EOF_CH       # the reference never unreads after EOF (scanner.go:79-228)
        ch = self.s[self.i]
        self.i += 1
        return ch

    def _unread(self):
        self.i -= 1

    def scan(self):
        ch = self._read()
        if _is_ws(ch):
            self._unread()
            return self._scan_ws()
        if ch == '"':
            self._unread()
            return self._scan_keyword(False)
        if _is_letter(ch):
            self._unread()
            return self._scan_operators()
        if ch == "(":
            return OPPAR, "("
        if ch == ")":
            return CLPAR, ")"
        if ch == _EOF_CH:
            return EOF, ""
        raise DslError("illegal char was found %s" % ch)

    def _scan_ws(self):
        buf = [self._read()]
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                break
            if not _is_ws(ch):
                self._unread()
                break
            buf.append(ch)
        return WS, "".join(buf)

    def _scan_operators(self):
        ch = self._read()
        if not _is_letter(ch):
            raise DslError("fail to scan operator: expected letter but found %s" % ch)
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
            return AND, lit
        if up == "OR":
            return OR, lit
        if up == "NOT":
            return NOT, lit
        if up == "INORD":
            return INORD, lit
        if up == "R":
            return self._scan_keyword(True)
        raise DslError("failed to scan operator: unexpected operator '%s' found" % lit)

    def _scan_keyword(self, is_regex):
        ch = self._read()
        scan_type = "regex" if is_regex else "keyword"
        if ch != '"':
            raise DslError('fail to scan %s: expected " but found %s' % (scan_type, ch))
        buf = []
        while True:
            ch = self._read()
            if ch == _EOF_CH:
                raise DslError('fail to scan %s: expected " but found EOF' % scan_type)
            if ch == "\\":
                sc = self._read()
                if sc == "\\":
                    buf.append(sc)
                elif sc == "n":
                    buf.append("\n")
                elif sc == "r":
                    buf.append("\r")
                elif sc == "t":
                    buf.append("\t")
                elif sc == '"':
                    buf.append(sc)
                else:
                    raise DslError("fail to scan %s: invalid escaped char %s" % (scan_type, sc))
            elif ch == '"':
                break
            else:
                buf.append(ch)
        return (REGEX if is_regex else KEYWORD), "".join(buf)


class Expression:
    """dsl/expression.go:42-48."""
    __slots__ = ("LExpr", "RExpr", "Type", "Literal", "Inord")

    def __init__(self, Type=UNSET_EXPR, LExpr=None, RExpr=None, Literal="", Inord=False):
        self.Type, self.LExpr, self.RExpr, self.Literal, self.Inord = Type, LExpr, RExpr, Literal, Inord

    def to_obj(self):
        d = {"Type": EXPR_NAMES[self.Type]}
        if self.Literal:
            d["Literal"] = self.Literal
        if self.Inord:
            d["Inord"] = True
        if self.LExpr is not None:
            d["LExpr"] = self.LExpr.to_obj()
        if self.RExpr is not None:
            d["RExpr"] = self.RExpr.to_obj()
        return d


class Parser:
    """dsl/parser.go:11-315."""

    def __init__(self, text, case_sensitive):
        self.s = Scanner(text)
        self.buf_tok, self.buf_lit, self.unscanned = ILLEGAL, "", False
        self.keywords, self.regexes = {}, {}       # insertion-ordered sets
        self.par_count = 0
        self.case_sensitive = case_sensitive
        self.inord = False

    # parser.go:254-288
    def _scan(self):
        if self.unscanned:
            self.unscanned = False
            return self.buf_tok, self.buf_lit
        tok, lit = self.s.scan()
        self.buf_tok, self.buf_lit = tok, lit
        return tok, lit

    def _unscan(self):
        self.unscanned = True

    def _scan_ignore_ws(self):
        tok, lit = self._scan()
        if tok == WS:
            tok, lit = self._scan()
        return tok, lit

    def parse(self):
        return self._parse()

    def _attach(self, exp, new):
        if exp.LExpr is None:
            exp.LExpr = new
        else:
            exp.RExpr = new

    # parser.go:58-216
    def _parse(self):
        exp = Expression(Inord=self.inord)
        while True:
            tok, lit = self._scan_ignore_ws()
            if tok == OPPAR:
                self._attach(exp, self._handle_open_par())
            elif tok in (KEYWORD, REGEX):
                if not self.case_sensitive:
                    lit = lit.lower()
                self._attach(exp, Expression(UNIT_EXPR, Literal=lit, Inord=self.inord))
                self._add_literal(tok, lit)
            elif tok == AND:
                exp = self._handle_dual_op(exp, AND_EXPR)
            elif tok == OR:
                exp = self._handle_dual_op(exp, OR_EXPR)
            elif tok == NOT:
                if self.inord:
                    raise DslError("invalid expression: INORD operator must not contain NOT operator")
                ntok, nlit = self._scan_ignore_ws()
                not_exp = Expression(NOT_EXPR)
                if ntok in (KEYWORD, REGEX):
                    if not self.case_sensitive:
                        nlit = nlit.lower()
                    not_exp.RExpr = Expression(UNIT_EXPR, Literal=nlit)
                    self._add_literal(ntok, nlit)
                elif ntok == OPPAR:
                    not_exp.RExpr = self._handle_open_par()
                else:
                    raise DslError("invalid expression: Unexpected token '%s' after NOT" % TOKEN_NAMES[ntok])
                self._attach(exp, not_exp)
            elif tok == INORD:
                if self.inord:
                    raise DslError("invalid expression: INORD operator must not contain INORD operator")
                ntok, _ = self._scan_ignore_ws()
                inord_exp = Expression(INORD_EXPR)
                if ntok != OPPAR:
                    raise DslError("invalid expression: Unexpected token '%s' after INORD" % TOKEN_NAMES[ntok])
                self.inord = True
                new = self._handle_open_par()
                self.inord = False
                inord_exp.RExpr = new
                self._attach(exp, inord_exp)
            elif tok in (CLPAR, EOF):
                if tok == CLPAR:
                    self.par_count -= 1
                if self.par_count < 0:
                    raise DslError("invalid expression: unexpected EOF found. Extra closing parentheses: %d"
                                   % (-self.par_count))
                final = exp
                if exp.Type == UNSET_EXPR:
                    if exp.RExpr is not None:
                        final = exp.RExpr
                    elif exp.LExpr is not None:
                        final = exp.LExpr
                    else:
                        raise DslError("invalid expression: unexpected EOF found")
                if final.Type in (AND_EXPR, OR_EXPR) and final.RExpr is None:
                    raise DslError("invalid expression: incomplete expression %s" % EXPR_NAMES[final.Type])
                return final
            else:
                raise DslError("invalid expression: Unexpected operator was found (%d = '%s')" % (tok, lit))

    # parser.go:220-251
    def _handle_dual_op(self, exp, exp_type):
        if exp.LExpr is None:
            raise DslError("invalid expression: no left expression was found for %s" % EXPR_NAMES[exp_type])
        if exp.RExpr is None:
            exp.Type = exp_type
            return exp
        exp = Expression(exp_type, LExpr=exp, Inord=self.inord)
        ntok, _ = self._scan_ignore_ws()
        if ntok == OPPAR:
            exp.RExpr = self._handle_open_par()
        else:
            self._unscan()
        return exp

    # parser.go:291-302
    def _handle_open_par(self):
        parlvl = self.par_count
        self.par_count += 1
        new = self._parse()
        if self.par_count != parlvl:
            raise DslError("invalid expression: Unexpected '('")
        return new

    # parser.go:305-315
    def _add_literal(self, tok, lit):
        if tok == REGEX:
            self.regexes[lit] = None
        elif tok == KEYWORD:
            self.keywords[lit] = None
        else:
            raise DslError("expected REGEX or KEYWORD tokens type to add literal to set but received: %s"
                           % TOKEN_NAMES[tok])


def parse(text, case_sensitive=True):
    """-> (Expression, keywords list, regexes list); raises DslError with the reference's message."""
    p = Parser(text, case_sensitive)
    e = p.parse()
    return e, list(p.keywords), list(p.regexes)


def flatten_forest(exprs):
    """Expression trees -> (node_table [n,5] int32 rows (type,l,r,lit,inord), roots, literals) for
    orc_set_expressions."""
    nodes, roots, lits, lit_ix = [], [], [], {}

    def lit_id(s):
        if s not in lit_ix:
            lit_ix[s] = len(lits)
            lits.append(s)
        return lit_ix[s]

    def walk(e):
        if e is None:
            return -1
        # iterative-safe for deep left chains: recurse on the right, loop on the left
        stack, order = [e], []
        while stack:
            n = stack.pop()
            order.append(n)
            if n.LExpr is not None:
                stack.append(n.LExpr)
            if n.RExpr is not None:
                stack.append(n.RExpr)
        ids = {}
        for n in reversed(order):       # children before parents
            ids[id(n)] = len(nodes)
            nodes.append([n.Type,
                          ids[id(n.LExpr)] if n.LExpr is not None else -1,
                          ids[id(n.RExpr)] if n.RExpr is not None else -1,
                          lit_id(n.Literal) if n.Type == UNIT_EXPR else -1,
                          1 if n.Inord else 0])
        return ids[id(e)]

    for e in exprs:
        roots.append(walk(e))
    return nodes, roots, lits
