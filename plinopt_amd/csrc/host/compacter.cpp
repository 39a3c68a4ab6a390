// ===========================================================================
// bin/compacter -- post-pass on the straight-line program of the single winner
// (reference src/compacter.cpp:27-68 `Compacter`, include/plinopt_programs.inl:
// 1157-1455 `variablesTrimer`; every reference pipeline reads
// `optimizer | compacter -s | SLPchecker`, bin/FDT.sh:58, Makefile:79-80).
// Same contract, this build's own rewriting (tokens, not the reference's text
// surgery): until nothing changes (or -O # rounds)
//   * no-ops and dead temporaries are removed,
//   * a variable that is only a copy of another one (`t3:=i3;`, `t9:=r4;`) is
//     replaced by it everywhere,
//   * with -s (default; -n switches it off) a temporary used exactly once is
//     written in place of its use, its top-level signs folded into the use's sign
//     and parentheses added only where a product or a quotient needs them,
//   * a leading minus is rotated behind a positive term (`x:=-a+b;` -> `x:=b-a;`).
// None of the rewrites adds a counted operation (`lineOperations`,
// plinopt_programs.inl:116-133: a sign after `:=` or `(` is not an addition), and
// the program computes the same outputs; the statistics line is the reference's
// ("# N elements instead of M").  Host only: it runs once, on one program.
// Usage: compacter [-s|-n] [-O #] [stdin|file.slp]
// ===========================================================================
#include "plo_compact.hpp"

using namespace plo;
using namespace plo::compact;

int main(int argc, char **argv)
{
    bool simplSingle = true; std::string filename; size_t numloops = 0; char ouv = 'o';
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") {
            std::clog << "Usage: " << argv[0] << " [-s/-n] [-O #] [stdin|file.prg]\n"
                      << "  -s/-n: replace/not-replace singly used variables\n"
                      << "  -O #: number of trim loops (default until stable)\n";
            exit(-1);
        } else if (a == "-s") simplSingle = true;
        else if (a == "-n" || a == "-ns") simplSingle = false;
        else if (a == "-O" && i + 1 < argc) numloops = strtoull(argv[++i], nullptr, 10);
        else filename = a;
    }
    try {
        std::vector<Line> P;
        if (filename.empty()) P = parse(std::cin);
        else { std::ifstream in(filename); if (!in) return -1; P = parse(in); }
        const size_t PVs = prog_size(P);
        std::clog << std::string(40, '#') << std::endl;
        compact_program(P, simplSingle, letter_outputs(ouv), numloops);
        print(std::cout, P);
        std::clog << "# \033[1;32m" << prog_size(P) << "\telements\tinstead of " << PVs << "\033[0m" << std::endl;
        std::clog << std::string(40, '#') << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return 3;
    }
    return 0;
}
