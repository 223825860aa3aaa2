// Meshlab filter plugin "Global registration" on the MI355X facade.  Same plugin class and MeshFilterInterface overrides
// as the reference's demos/MeshlabPlugin/filter_globalregistration/globalregistration.h:39-59, so the plugin project file
// of the reference builds it unchanged.
#ifndef S4P_MESHLAB_GLOBALREGISTRATION_H_
#define S4P_MESHLAB_GLOBALREGISTRATION_H_

#include <common/interfaces.h>

class GlobalRegistrationPlugin : public QObject, public MeshFilterInterface {
  Q_OBJECT
  MESHLAB_PLUGIN_IID_EXPORTER(MESH_FILTER_INTERFACE_IID)
  Q_INTERFACES(MeshFilterInterface)

 public:
  enum { FP_GLOBAL_REGISTRATION };

  GlobalRegistrationPlugin();

  virtual QString pluginName(void) const { return "GlobalRegistrationPlugin"; }

  QString filterName(FilterIDType filter) const;
  QString filterInfo(FilterIDType filter) const;
  void initParameterSet(QAction*, MeshDocument& /*md*/, RichParameterSet& /*parent*/);
  bool applyFilter(QAction* filter, MeshDocument& md, RichParameterSet& /*parent*/, vcg::CallBackPos* cb);
  int postCondition(QAction*) const { return MeshModel::MM_VERTCOORD; }
  FilterClass getClass(QAction* a);
  FILTER_ARITY filterArity(QAction*) const { return SINGLE_MESH; }
};

#endif  // S4P_MESHLAB_GLOBALREGISTRATION_H_
