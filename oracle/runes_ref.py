"""Test infrastructure: Go's string -> []rune conversion, restated (the reference's AnknownEngine searches []rune(text) and
reports rune offsets: finder/substringEngine.go:44-53).  `range` over a string decodes with utf8.DecodeRuneInString (Go
standard library, unicode/utf8 -- not part of /root/reference; restated from its published algorithm: first-byte table,
accept ranges for the second byte, continuation bytes 80..BF, every error is one U+FFFD of width 1)."""


def decode_rune_len(b, i):
    """width of the rune that starts at b[i] (1 for ASCII and for every error)"""
    n = len(b)
    b0 = b[i]
    if b0 < 0xC2 or b0 > 0xF4:
        return 1                       # ASCII; 80..C1 and F5..FF are invalid first bytes
    size = 2 if b0 < 0xE0 else 3 if b0 < 0xF0 else 4
    if i + size > n:
        return 1                       # short: RuneError, width 1
    lo, hi = 0x80, 0xBF                # acceptRanges
    if b0 == 0xE0:
        lo = 0xA0
    elif b0 == 0xED:
        hi = 0x9F
    elif b0 == 0xF0:
        lo = 0x90
    elif b0 == 0xF4:
        hi = 0x8F
    if not lo <= b[i + 1] <= hi:
        return 1
    for k in range(2, size):
        if not 0x80 <= b[i + k] <= 0xBF:
            return 1
    return size


def rune_index_table(b):
    """t[p] = number of runes of []rune(b) that start in front of byte p, for p in 0..len(b)"""
    t = [0] * (len(b) + 1)
    i = r = 0
    while i < len(b):
        w = decode_rune_len(b, i)
        for k in range(w):
            t[i + k] = r + (1 if k else 0)      # (a position inside a rune: that rune has started)
        i += w
        r += 1
    t[len(b)] = r
    return t
