// xstd::output_iterator_traits -- drop-in for include/pcp/traits/output_iterator_traits.hpp:21-54: the value type an
// output iterator writes.  std::iterator_traits gives void for the standard insert iterators; each of them names its
// container (container_type), which is where the value type is taken from here.
#ifndef PCP_TRAITS_OUTPUT_ITERATOR_TRAITS_HPP
#define PCP_TRAITS_OUTPUT_ITERATOR_TRAITS_HPP

#include <iterator>
#include <type_traits>

namespace xstd {

namespace detail {
template <class It, class = void>
struct written_type
{
    using type = typename std::iterator_traits<It>::value_type;
};
template <class It>
struct written_type<It, std::void_t<typename It::container_type>>
{
    using type = typename It::container_type::value_type;
};
} // namespace detail

template <class It>
struct output_iterator_traits
{
    using value_type = typename detail::written_type<It>::type;
};

} // namespace xstd

#endif
