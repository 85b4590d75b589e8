// Stand-in for <opm/models/utils/propertysystem.hh> of opm-models (NOT in the reference tree: SURVEY.md section 8c), own code, so that
// host/HipLinearizer.hpp compiles here the way it compiles inside Flow.  Only what the linearizer names: the property tags it
// reads and GetPropType / getPropValue.  In an opm-simulators tree the real header takes this one's place (same include path).
#pragma once
namespace Opm {
namespace Properties {
template <class TypeTag, class MyTypeTag> struct Simulator;
template <class TypeTag, class MyTypeTag> struct SparseMatrixAdapter;
template <class TypeTag, class MyTypeTag> struct GlobalEqVector;
template <class TypeTag, class MyTypeTag> struct SolutionVector;
template <class TypeTag, class MyTypeTag> struct RateVector;
template <class TypeTag, class MyTypeTag> struct Linearizer;
}  // namespace Properties
template <class TypeTag, template <class, class> class Property>
using GetPropType = typename Property<TypeTag, TypeTag>::type;
}  // namespace Opm
