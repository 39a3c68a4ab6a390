#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- never imported, linked or executed by the product.

Literal restatement of the reference's program compacter (`bin/compacter`):
    Compacter          /root/reference/src/compacter.cpp:27-68
    programParser      /root/reference/include/plinopt_programs.inl:618-686
    variablesTrimer    /root/reference/include/plinopt_programs.inl:1157-1455
and of every helper they call (each function cites its lines).  A program is what the
reference holds: a list of lines, each a list of words (`VProgram_t`,
plinopt_programs.h:32); the restatement keeps the reference's text surgery word for
word -- `std::rotate` + `erase`, iterator arithmetic, the order of the passes, the
quirks (the numeric right-hand side of an output that is back-substituted as if it were
a variable, `:1215-1230`; `size_t` depths that wrap below zero) -- so that the product's
`bin/compacter`, an engine of its own, can be held to it line by line.

Third-party behaviour (Givaro, absent from the tree): `Givaro::Rational(const char*)`
on "n" / "n/d" and its printing as "n" / "n/d" in lowest terms are mathematically
determined and restated with `fractions.Fraction`; what the constructor does with a
word that is not a number (`:1355-1361` builds one from whatever follows a `*` or `/`
when two of them are a word apart) is NOT pinned: `NotPinned` is raised and the tests
skip such an input.  Reading an element the reference would read out of range
(undefined behaviour there) raises IndexError here.

Pinned by a test the reference holds: `bin/GDT.sh:41-65` (`make opcheck`) asserts that
for every stored `data/*.slp` the operations `SLPchecker` counts (lineOperations,
`:116-133`) in `compacter f` equal the `sed` count of `bin/OpCount.sh:18` on `f`;
`tests/test_compacter.py` runs that on this restatement and on the product.

Command line (the reference's): plo_compact_oracle.py [-s|-n] [-O #] [file]
"""
import sys
from fractions import Fraction


class NotPinned(Exception):
    pass


def ch0(s):
    """s[0] of a std::string: the terminator for an empty one."""
    return s[0] if s else "\0"


# ---- small predicates, plinopt_programs.inl:32-107 ---------------------------------------
def is_natural(s):                     # :32-35
    return all(c.isdigit() and c.isascii() for c in s)


def idempots(s):                       # :77-78
    return len(s) == 4 and s[0] == s[2]


def is_add_sub(s):                     # :90-91
    return s == "+" or s == "-"


def is_mul_div(s):                     # :93-94
    return s == "*" or s == "/"


def is_par_aff(s):                     # :96-97
    return s == "(" or s == ":="


def is_variable(s):                    # :99-100
    return not any(c in "+-*/;:=()" for c in s)


def swapsign(s):                       # :103-106
    return "-" if s == "+" else "+" if s == "-" else s


def prog_size(P):                      # :110-114
    return sum(len(line) for line in P) - 2 * len(P)


def line_operations(line):             # :116-133
    adds = muls = 0
    negator = False
    for word in line:
        if is_add_sub(word) and not negator:
            adds += 1
        elif is_mul_div(word):
            muls += 1
        negator = is_par_aff(word)
    return adds, muls


def prog_operations(P):                # :135-141
    a = m = 0
    for line in P:
        x, y = line_operations(line)
        a += x
        m += y
    return a, m


def unused_char(C, cstart="`"):        # :246-259 ('a'-1 is '`')
    if len(C) > 50:
        raise RuntimeError("not enough free single char variables.")
    t = ord(cstart)
    while True:
        t += 1
        if not (chr(t) in C or chr(t) == "c"):
            break
    if t > ord("z"):
        t = ord("A")
        while chr(t) in C:
            t += 1
    return chr(t)


def string_trimer(s):                  # plinopt_programs.h:130-139
    h = s.find("#")
    if h >= 0:
        s = s[:h]
    return s.rstrip(" \t\n\v\f\r")


def _find_first_of(s, chars, start):
    for k in range(start, len(s)):
        if s[k] in chars:
            return k
    return -1


def program_parser(text):              # :618-686
    P = []
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                    # std::getline yields nothing after the last newline
    for line in lines:
        line = string_trimer(line)
        post = line.find(":=")
        if post < 0:
            continue
        wv = [line[:post], ":="]
        prev = post + 2
        while True:
            pos = _find_first_of(line, ")(+-*/;", prev)
            if pos < 0:
                break
            delim = line[pos]
            if pos > prev:
                node = line[prev:pos]
                if delim == "/" and is_natural(node):
                    prev = pos + 1
                    pos = _find_first_of(line, "()+-*;", prev)
                    if pos < 0:
                        raise IndexError("rational without a delimiter behind it (:651 reads line.substr(npos,1))")
                    node = node + delim + line[prev:pos]
                    delim = line[pos]
                wv.append(node)
            wv.append(delim)
            prev = pos + 1
        if prev < len(line):
            wv.append(line[prev:])
        P.append(wv)
    return P


def program_text(P):                   # operator<<, :40-47
    return "".join("".join(line) + "\n" for line in P if len(line) > 0)


def _rotate(v, first, middle, last):
    """std::rotate on v[first:last]: v[middle] becomes the first element."""
    v[first:last] = v[middle:last] + v[first:middle]


# ---- sign moves, :693-966 -----------------------------------------------------------------
def rotate_minus(line):                # :693-723 (RANDOM_TIES is not defined by the Makefile)
    if line[2] == "-":
        depth = 0
        for i in range(3, len(line)):
            if line[i] == "(":
                depth += 1
            if line[i] == ")":
                depth -= 1
            if depth == 0:
                if line[i] == "+":
                    _rotate(line, 2, i, len(line) - 1)
                    del line[2]
                    return True
    return False


def rotate_group_minus(block, offset=0):     # :726-753
    startb = offset + 1
    if block[offset] == "(" and block[startb] == "-":
        depth = 0
        foundplus = False
        lastplus = closp = None
        for it in range(startb + 1, len(block)):
            if block[it] == "(":
                depth += 1
            if block[it] == ")":
                depth -= 1
            if depth < 0:
                closp = it
                break
            if depth == 0 and block[it] == "+":
                lastplus = it
                foundplus = True
        if foundplus:
            if closp is None:
                raise IndexError("rotateGroupMinus: group is not closed (:746 uses an unset iterator)")
            _rotate(block, startb, lastplus, closp)
            del block[startb]
            return True
    return False


def negate_line(line):                 # :757-789
    depth = 0
    var = [line[0], line[1]]
    j = 2
    if line[j] != "-":
        var.append("-")
        j += 1
    var.extend(line[2:])
    while j < len(var):
        if var[j] == "(":
            depth += 1
        if var[j] == ")":
            depth -= 1
        if depth == 0:
            if var[j] == "-":
                if is_par_aff(var[j - 1]):
                    del var[j]         # rotate(j, j+1, end); pop_back
                    j -= 1
                else:
                    var[j] = "+"
            elif var[j] == "+":
                var[j] = "-"
        j += 1
    return var


def negating_variable(oldline, variable):    # :794-821; returns (linmod, line)
    linmod = False
    line = list(oldline)
    j = 2
    while j < len(line):
        if line[j] == variable:
            if line[j - 1] == "+":
                line[j - 1] = "-"
            elif line[j - 1] == "-":
                if is_par_aff(line[j - 2]):
                    del line[j - 1]    # rotate(j-1, j, end); pop_back -- j is NOT moved back
                else:
                    line[j - 1] = "+"
            elif is_par_aff(line[j - 1]):
                line.insert(j, "-")    # push_back("-"); rotate(j, end-1, end)
                j += 1
            linmod = True
        j += 1
    return linmod, line


def min_line(vP, index, inchar, outchar, force=False):     # :828-883; varline is vP[index]
    varline = vP[index]
    if ch0(varline[0]) == outchar:
        rotate_minus(varline)
        if varline[2] == "-":
            j = 3
            while j < len(varline):
                variable = varline[j]
                nP = [list(l) for l in vP]
                if is_variable(variable) and ch0(variable) != inchar and ch0(variable) != outchar:
                    cm = 1
                    for k in range(index - 1, -1, -1):
                        if nP[k][0] == variable:
                            bkm = 1 if nP[k][2] == "-" else 0
                            refline = negate_line(nP[k])
                            rotate_minus(refline)
                            dkm = (1 if refline[2] == "-" else 0) - bkm
                            if force or dkm <= 0:
                                cm += dkm
                            else:
                                break
                            nP[k] = refline
                            for l in range(k + 1, len(nP)):
                                blm = 1 if nP[l][2] == "-" else 0
                                _, negline = negating_variable(nP[l], variable)
                                rotate_minus(negline)
                                nP[l] = negline
                                dlm = (1 if nP[l][2] == "-" else 0) - blm
                                cm += dlm
                                if nP[l][0] == variable:
                                    break
                    if force or cm <= 0:
                        vP[:] = nP
                        return True
                j += 1
    return False


def ending_minus(vP, inchar, outchar, force=False):        # :885-899
    if not vP:
        return 0
    rm = 0
    for i in range(len(vP) - 1, -1, -1):
        if min_line(vP, i, inchar, outchar):
            rm += 1
        elif force:
            min_line(vP, i, inchar, outchar, True)
    return rm


def useless_plus(line):                # :905-911
    if len(line) > 2 and line[1] == ":=" and line[2] == "+":
        del line[2]
        return True
    return False


def count_minus(P):                    # :916-920
    return sum(1 for line in P if line[2] == "-")


def swap_minus(P, outchar):            # :926-966
    for line in P:
        rotate_minus(line)
    pm = count_minus(P)
    if pm == 0:
        return 0
    i = 0
    while i < len(P):
        if P[i][2] == "-" and ch0(P[i][0]) != outchar:
            mP = {}
            mP[i] = negate_line(P[i])
            variable = mP[i][0]
            while True:
                i += 1
                if not i < len(P):
                    break
                linmod, line = negating_variable(P[i], variable)
                if linmod:
                    rotate_minus(line)
                    mP[i] = line
                if P[i][0] == variable:
                    break
            cm = pm
            for k in sorted(mP):
                if P[k][2] == "-" and mP[k][2] != "-":
                    cm -= 1
                if mP[k][2] == "-" and P[k][2] != "-":
                    cm += 1
            if cm < pm:
                for k in sorted(mP):
                    P[k] = mP[k]
                pm = cm
        i += 1
    return pm


# ---- parentheses, :969-1142 ---------------------------------------------------------------
def _find_first_of_words(v, start, end, words):
    for k in range(start, end):
        if v[k] in words:
            return k
    return end


def end_group(v, start, end):          # :969-986
    if v[start] == "(":
        depth = 1
        closp = start
        while True:
            closp = _find_first_of_words(v, closp + 1, end, ("(", ")"))
            if closp >= end:
                raise IndexError("endGroup: unbalanced parenthesis (:977 dereferences end)")
            if v[closp] == ")":
                depth -= 1
            else:
                depth += 1
            if not depth > 0:
                break
        return _find_first_of_words(v, closp + 1, end, ("+", "-", ")", ";"))
    return _find_first_of_words(v, start, end, ("+", "-"))


def parenthesis_minus_line(line):      # :994-1064
    swapped = False
    newline = []
    startl = 0
    while startl != len(line):
        openp = _find_first_of_words(line, startl, len(line), ("(",))
        if openp != len(line):
            closp = end_group(line, openp, len(line))
            minusign = 0
            newgroup = []
            if line[openp - 1] == "-":
                newgroup = line[openp - 1:closp]
                newgroup[0] = "+"
                nosign = 0
                it = 2
                while True:
                    if newgroup[it] == ")":
                        break
                    elif newgroup[it] == "-":
                        newgroup[it] = "+"
                        it += 1
                        minusign += 1
                    elif newgroup[it] == "+":
                        newgroup[it] = "-"
                        it += 1
                    else:
                        nosign += 1
                    it = end_group(newgroup, it, len(newgroup))
                    if it == len(newgroup):
                        break
                if nosign > 0:
                    newgroup.insert(2, "-")
                if newgroup[2] == "+":
                    del newgroup[2]
                rotate_group_minus(newgroup)
                rotate_group_minus(newgroup, 1)
            if minusign > 0:
                newline += line[startl:openp - 1]
                newline += newgroup
                swapped = True
            else:
                newline += line[startl:closp]
            startl = closp - 1
        else:
            newline += line[startl:]
            break
        startl += 1
    useless_plus(newline)
    if swapped:
        line[:] = newline
    return line


def swap_parenthesis_minus(v):         # :1075-1142, on the words of [start, endl)
    newline = []
    startl = 0
    endl = len(v)
    while startl != endl:
        openp = _find_first_of_words(v, startl, endl, ("(",))
        newline += v[startl:openp]
        if openp == endl:
            break
        closp = end_group(v, openp, endl)
        offset = 0
        if openp != startl:
            offset += 1
            newline.pop()
        newgroup = v[openp - offset:closp]
        offset += 1
        if openp != startl and newgroup[offset] == "-":
            swap = False
            if newgroup[0] == "+":
                newgroup[0] = "-"
                swap = True
            elif newgroup[0] == "-":
                newgroup[0] = "+"
                swap = True
            else:
                rotate_group_minus(newgroup, offset - 1)
            if swap:
                newgroup = negate_line(newgroup)
        elif newgroup[-1] == ")":
            if newgroup[1] == "(":
                if newgroup[0] == "-":
                    newgroup = negate_line(newgroup)
                    rem_par = 1
                    if newgroup[2] == "+" or newgroup[2] == "-":
                        rem_par -= 1
                    newgroup.pop()
                    del newgroup[rem_par:2]
                if newgroup[0] == "+" or newgroup[0] == ":=":
                    newgroup.pop()
                    del newgroup[1:2]
        if len(newgroup) < offset:
            raise IndexError("swapParenthesisMinus: group shorter than its offset (:1131)")
        newline += newgroup[:offset]
        newline += swap_parenthesis_minus(newgroup[offset:])
        startl = closp
    useless_plus(newline)
    return newline


# ---- Givaro::Rational on words -------------------------------------------------------------
def _rational(word):
    parts = word.split("/")
    if not (1 <= len(parts) <= 2) or not all(p and is_natural(p) for p in parts):
        raise NotPinned("Givaro::Rational(%r): not a number" % word)
    if len(parts) == 2 and int(parts[1]) == 0:
        raise NotPinned("Givaro::Rational(%r): zero denominator" % word)
    return Fraction(int(parts[0]), int(parts[1]) if len(parts) == 2 else 1)


# ---- variablesTrimer, :1157-1455 -----------------------------------------------------------
def variables_trimer(P, simpl_single=True, inchar="i", outchar="o"):
    vars_char = set(ch0(word) for line in P for word in line)
    freechar = unused_char(vars_char)
    tmpchar = unused_char(vars_char, freechar)

    # [1] output variables only at the end of the program, :1169-1187
    OutP = []
    for li in range(len(P)):
        outvar = P[li][0]
        if ch0(outvar) == outchar:
            repvar = freechar + outvar[1:]
            for nx in range(li, len(P)):
                nl = P[nx]
                for w in range(len(nl)):
                    if nl[w] == outvar:
                        nl[w] = repvar
            OutP.append([outvar, ":=", repvar, ";"])
    P.extend(OutP)

    # [2] direct substitution of simple affectations, :1189-1206
    for li in range(len(P)):
        line = P[li]
        if len(line) == 4 and ch0(line[0]) != outchar:
            outvar, invar = line[0], line[2]
            for nx in range(li + 1, len(P)):
                nl = P[nx]
                for w in range(2, len(nl)):
                    if nl[w] == outvar:
                        nl[w] = invar
                if nl[0] == outvar:
                    break
            line[0] = invar

    P[:] = [l for l in P if not idempots(l)]           # :1210

    # [3] backward substitution of outputs, :1212-1232
    for li in range(len(P) - 1, -1, -1):
        line = P[li]
        if len(line) == 4 and line[2] != "0" and ch0(line[2]) != inchar and ch0(line[2]) != outchar:
            for nx in range(li - 1, -1, -1):
                nl = P[nx]
                if nl[0] == line[2]:
                    nl[0] = line[0]
                    line[2] = line[0]
                    break
                for w in range(2, len(nl)):
                    if nl[w] == line[2]:
                        nl[w] = line[0]

    P[:] = [l for l in P if not idempots(l)]           # :1236

    for line in P:                                       # :1240
        rotate_minus(line)
    ending_minus(P, inchar, outchar)                     # :1244
    if not simpl_single:
        return 0

    # [4] singly used variables, :1248-1407
    tmpnum = 0
    vars_set, vars_use = {}, {}
    for i in range(len(P)):
        line = P[i]
        prevar = line[0]
        if ch0(prevar) != outchar and line[0] in vars_set:
            tmpnum += 1
            line[0] = tmpchar + str(tmpnum)
            for j in range(i + 1, len(P)):
                nl = P[j]
                for w in range(len(nl)):
                    if nl[w] == prevar:
                        nl[w] = line[0]
        vars_set.setdefault(line[0], []).append(i)
        for word in line:
            if word in vars_set:
                vars_use.setdefault(word, []).append(i)

    ords_use = [(v, list(o)) for v, o in vars_use.items()]
    fronts = [o[0] for _, o in ords_use]
    assert len(set(fronts)) == len(fronts), "std::sort :1285 on equal keys is not pinned"
    ords_use.sort(key=lambda vo: vo[1][0])

    for variable, occ in ords_use:
        if ch0(variable) != outchar and len(occ) == 2:
            i, j = occ[0], occ[1]
            if i == j:
                raise IndexError("variable used in its own first assignment (:1291 aliases init and line)")
            init, line = P[i], P[j]
            rotate_minus(init)
            varloc = 2
            while varloc < len(line):
                if line[varloc] == variable:
                    break
                varloc += 1
            if varloc < len(line):
                if init[2] == "-" and is_add_sub(line[varloc - 1]):      # :1308-1312
                    line[varloc - 1] = swapsign(line[varloc - 1])
                    if useless_plus(line):
                        varloc -= 1
                    init[:] = negate_line(init)
                multimonomial = False                                      # :1315-1325
                depth = 0
                for it in range(3, len(init) - 1):
                    if init[it] == "(":
                        depth += 1
                    elif init[it] == ")":
                        depth -= 1
                    elif is_add_sub(init[it]) and depth == 0:
                        multimonomial = True
                        break
                replength = len(init) - 2
                tobeneg = line[varloc - 1] == "-"
                tobemul = is_mul_div(line[varloc + 1])
                if (not tobemul) and tobeneg and multimonomial:           # :1332-1350
                    init[:] = negate_line(init)
                    rotate_minus(init)
                    replength = len(init) - 2
                    if varloc == 3 or init[2] == "-":
                        varloc -= 1
                        del line[varloc]
                    else:
                        line[varloc - 1] = "+"
                        if useless_plus(line):
                            varloc -= 1
                    tobeneg = False
                line[varloc] = init[2]                                     # :1352-1358
                init[2] = ""                                               # moved-from word
                line[varloc + 1:varloc + 1] = init[3:len(init) - 1]
                if multimonomial and (tobemul or tobeneg):
                    line.insert(varloc, "(")
                    line.insert(varloc + replength, ")")
                pos = 0                                                    # :1361-1405
                while pos < len(line) - 4:
                    if is_mul_div(line[pos]) and is_mul_div(line[pos + 2]):
                        lcoeff = _rational(line[pos + 1])
                        if line[pos] == "/":
                            lcoeff = 1 / lcoeff
                        rmulti = _rational(line[pos + 3])
                        if line[pos + 2] == "/":
                            lcoeff /= rmulti
                        else:
                            lcoeff *= rmulti
                        if lcoeff.denominator == 1:
                            if lcoeff.numerator == 1:
                                del line[pos:pos + 4]
                            else:
                                line[pos] = "*"
                                line[pos + 1] = str(lcoeff.numerator)
                                del line[pos + 2:pos + 4]
                        else:
                            if lcoeff.numerator == 1:
                                line[pos] = "/"
                                line[pos + 1] = str(lcoeff.denominator)
                            else:
                                line[pos] = "*"
                                line[pos + 1] = "%d/%d" % (lcoeff.numerator, lcoeff.denominator)
                            del line[pos + 2:pos + 4]
                    pos += 1
                del init[:]                                                # :1407

    P[:] = [l for l in P if len(l) != 0]               # :1419

    for k in range(len(P)):                              # :1423-1432
        line = swap_parenthesis_minus(P[k])
        if line[1] == ":=" and line[2] == "(" and line[3] == "-":
            line.insert(2, "+")
            line = swap_parenthesis_minus(line)
        rotate_minus(line)
        P[k] = line

    if count_minus(P) > 0:                               # :1438-1441
        for line in P:
            parenthesis_minus_line(line)
        swap_minus(P, outchar)

    ending_minus(P, inchar, outchar, True)               # :1445
    for line in P:                                       # :1449
        rotate_minus(line)
    return tmpnum


def compacter(text, numloops=0, simpl_single=True, log=None):     # src/compacter.cpp:27-68
    P = program_parser(text)
    PVs = prog_size(P)
    if log is not None:
        log.append("#" * 40)
    variables_trimer(P, simpl_single)
    curr = prog_size(P)
    it = numloops
    while True:
        prev = curr
        variables_trimer(P, simpl_single)
        curr = prog_size(P)
        if log is not None:
            log.append("# %d\telements\tinstead of %d" % (curr, prev))
        it -= 1
        if not (curr < prev and it != 0):
            break
    if log is not None:
        log.append("# \033[1;32m%d\telements\tinstead of %d\033[0m" % (curr, PVs))
        log.append("#" * 40)
    return program_text(P)


def main(argv):
    simpl, filename, numloops = True, "", 0
    i = 1
    while i < len(argv):
        a = argv[i]
        if a == "-s":
            simpl = True
        elif a in ("-n", "-ns"):
            simpl = False
        elif a == "-O":
            i += 1
            numloops = int(argv[i])
        else:
            filename = a
        i += 1
    text = open(filename).read() if filename else sys.stdin.read()
    log = []
    out = compacter(text, numloops, simpl, log)
    sys.stdout.write(out)
    sys.stderr.write("\n".join(log) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
