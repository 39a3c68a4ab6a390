// ===========================================================================
// bin/SLPchecker -- rebuilds the matrix computed by an SLP and compares it with
// -M file.sms, over Q or modulo -q (reference src/SLPchecker.cpp:22-105: same
// flags, same SUCCESS/ERROR lines on stderr; without -M the matrix is printed
// in SMS format, which is how data/32x32x32_15096_L.sms is regenerated,
// reference Makefile:79-80).
// ===========================================================================
#include "plo_host.hpp"
#include "plo_trim.hpp"

using namespace plo;

template <class F> int check(const F &f, const std::string &prg, const std::string &mat)
{
    // as SLPbuilder (src/SLPchecker.cpp:27-45): the whole program is parsed and its operations are counted and printed
    // BEFORE the matrix is built -- bin/GDT.sh:44 reads these two lines whatever the evaluation says afterwards
    std::stringstream text;
    if (prg.empty()) text << std::cin.rdbuf();
    else { std::ifstream in(prg); if (!in) { std::cerr << "# ERROR: cannot open " << prg << std::endl; return 2; } text << in.rdbuf(); }
    std::pair<size_t, size_t> ops;
    { trim::Words W; std::istringstream in(text.str()); ops = trim::operations(trim::parse(in, W)); }
    std::clog << std::string(40, '#') << std::endl;
    std::clog << "# \033[1;32m" << ops.first << "\tadditions\033[0m" << std::endl;
    std::clog << "# \033[1;32m" << ops.second << "\tmultiplications\033[0m" << std::endl;
    std::clog << std::string(40, '#') << std::endl;
    SlpEval<F> ev(f);
    ev.run(text);
    if (mat.empty()) { auto A = ev.matrix(); write_sms(std::cout, f, A, std::is_same<F, QField>::value ? 'R' : 'M'); return 0; }
    std::ifstream mf(mat);
    if (!mf) { std::cerr << "# ERROR: cannot open " << mat << std::endl; return 2; }
    QMat MQ = read_sms(mf);
    auto nb = naive_ops(QField(), MQ);
    auto B = rebind(MQ, f);
    auto A = ev.matrix('o', B.rowdim(), B.coldim());
    size_t m = std::max(A.rowdim(), B.rowdim()), n = std::max(A.coldim(), B.coldim());
    if (same_matrix(f, A, B)) {
        std::clog << "# \033[1;32mSUCCESS: correct SLP for " << m << 'x' << n << " (" << nb.first << "+|" << nb.second << "x) : "
                  << ops.first << ',' << ops.second << " Matrix-Vector multiplication!\033[0m" << std::endl;
        return 0;
    }
    std::cerr << "# \033[1;31m****** ERROR, not a " << m << 'x' << n << " m-v algorithm******\033[0m" << std::endl;
    return 1;
}

int main(int argc, char **argv)
{
    std::string prg, mat; uint64_t q = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") { std::clog << "Usage: " << argv[0] << " [-q #] [-M file.sms] [stdin|file.slp]\n"; exit(-1); }
        else if (a == "-M" && i + 1 < argc) mat = argv[++i];
        else if (a == "-q" && i + 1 < argc) q = strtoull(argv[++i], nullptr, 10);
        else prg = a;
    }
    try {
        if (q >= (1ull << 31)) return check(Zp64Field(q), prg, mat);
        if (q) return check(ZpField((uint32_t)q), prg, mat);
        return check(QField(), prg, mat);
    } catch (const std::exception &e) {
        std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return 3;
    }
}
